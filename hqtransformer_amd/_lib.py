"""ctypes binding of libhqt.so (C ABI: include/hqt.h) and its in-tree build recipe.

There is deliberately no CPU fallback: if the shared library is missing or cannot be loaded, every
entry point of the package raises ``HqtLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libhqt.so')
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['engine.hip', 'kernels.hip', 'fast_kernels.hip', 'persist.hip', 'tile_gemm.hip', 'exact_gemm.hip', 'mfma_gemm.hip', 'split_conv.hip', 'split_stream_conv.hip']
ABI_VERSION = 7

PRECISION_EXACT, PRECISION_FAST, PRECISION_SPLIT = 0, 1, 2
PRECISIONS = {'exact': PRECISION_EXACT, 'fast': PRECISION_FAST, 'split': PRECISION_SPLIT}
POLICY_LATENCY, POLICY_THROUGHPUT = 0, 1
SWITCH_PERSIST, SWITCH_SINGLE_KEY, SWITCH_PERSIST_FAULT, SWITCH_SPLIT_KSLICES = 0, 1, 2, 3
LAYOUT_FAST, LAYOUT_EXACT, LAYOUT_SPLIT, LAYOUT_ALL = 1, 2, 4, 7


class HqtLibraryError(RuntimeError):
    pass


class HqtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f'libhqt error {code}: {msg}')
        self.code = code


class hqt_config(C.Structure):
    _fields_ = [
        ('abi_version', C.c_int32),
        ('has_stage2', C.c_int32),
        ('embed_dim', C.c_int32), ('n_layers', C.c_int32), ('n_heads', C.c_int32), ('n_layers_depth', C.c_int32),
        ('vocab_top', C.c_int32), ('vocab_bot', C.c_int32), ('vocab_txt', C.c_int32),
        ('ctx_len_img', C.c_int32), ('ctx_len_txt', C.c_int32), ('n_classes', C.c_int32),
        ('cond_type', C.c_int32), ('embedding_type', C.c_int32), ('gelu_approx', C.c_int32),
        ('has_stage1', C.c_int32),
        ('s1_ch', C.c_int32), ('s1_n_mult', C.c_int32), ('s1_ch_mult', C.c_int32 * 8),
        ('s1_num_res_blocks', C.c_int32),
        ('s1_n_attn_res', C.c_int32), ('s1_attn_res', C.c_int32 * 4),
        ('s1_resolution', C.c_int32), ('s1_z_channels', C.c_int32), ('s1_embed_dim', C.c_int32),
        ('s1_n_embed', C.c_int32), ('s1_out_ch', C.c_int32),
        ('s1_use_init_downsample', C.c_int32), ('s1_use_mid_block', C.c_int32), ('s1_use_attn', C.c_int32),
        ('max_batch', C.c_int32), ('max_steps', C.c_int32),
        ('code_levels', C.c_int32),
        ('depth_decoding', C.c_int32),
        ('ar_layouts', C.c_int32),
    ]


class hqt_sample_opts(C.Structure):
    _fields_ = [
        ('precision', C.c_int32), ('n_steps', C.c_int32),
        ('top_k_top', C.c_int32), ('top_k_bot', C.c_int32),
        ('top_p_top', C.c_float), ('top_p_bot', C.c_float),
        ('temperature_top', C.c_float), ('temperature_bot', C.c_float),
        ('seed', C.c_uint64), ('sample_offset', C.c_int64),
        ('use_graph', C.c_int32),
        ('row_seeds', C.c_void_p), ('row_offsets', C.c_void_p),
    ]


class hqt_sample_opts_l3(C.Structure):
    _fields_ = [
        ('precision', C.c_int32), ('n_steps', C.c_int32),
        ('top_k', C.c_int32 * 3), ('top_p', C.c_float * 3), ('temperature', C.c_float * 3),
        ('seed', C.c_uint64), ('sample_offset', C.c_int64),
        ('use_graph', C.c_int32),
        ('row_seeds', C.c_void_p), ('row_offsets', C.c_void_p),
    ]


class hqt_encode_out(C.Structure):
    _fields_ = [
        ('codes', C.c_void_p * 3), ('quant', C.c_void_p * 3), ('resid', C.c_void_p * 3),
        ('recon', C.c_void_p), ('diff', C.c_void_p),
    ]


# every symbol include/hqt.h declares: name -> (restype, argtypes)
_VP, _I64P, _F32P = C.c_void_p, C.c_void_p, C.c_void_p
SYMBOLS = {
    'hqt_abi_version': (C.c_int, []),
    'hqt_last_error': (C.c_char_p, []),
    'hqt_create': (C.c_int, [C.POINTER(hqt_config), C.c_int, C.POINTER(_VP)]),
    'hqt_set_weight': (C.c_int, [_VP, C.c_char_p, _VP, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    'hqt_finalize_weights': (C.c_int, [_VP]),
    'hqt_clone': (C.c_int, [_VP, C.POINTER(_VP)]),
    'hqt_set_policy': (C.c_int, [_VP, C.c_int]),
    'hqt_set_switch': (C.c_int, [_VP, C.c_int, C.c_int]),
    'hqt_sample': (C.c_int, [_VP, C.c_int, _I64P, C.POINTER(hqt_sample_opts), _F32P, _I64P, _I64P, _F32P, _I64P, _I64P, _VP]),
    'hqt_decode': (C.c_int, [_VP, C.c_int, _I64P, _I64P, _F32P, C.c_int, C.c_int, _VP]),
    'hqt_decode_seq': (C.c_int, [_VP, C.c_int, _I64P, _I64P, _F32P, C.c_int, C.c_int, _VP]),
    'hqt_sample_l3': (C.c_int, [_VP, C.c_int, _I64P, C.POINTER(hqt_sample_opts_l3), _F32P, _I64P, _I64P, _I64P, _F32P, _I64P, _I64P, _I64P, _VP]),
    'hqt_decode_l3': (C.c_int, [_VP, C.c_int, _I64P, _I64P, _I64P, _F32P, C.c_int, C.c_int, _VP]),
    'hqt_decode_seq_l3': (C.c_int, [_VP, C.c_int, _I64P, _I64P, _I64P, _F32P, C.c_int, C.c_int, _VP]),
    'hqt_encode': (C.c_int, [_VP, C.c_int, _F32P, C.c_int, C.POINTER(hqt_encode_out), _VP]),
    'hqt_has_encoder': (C.c_int, [_VP]),
    'hqt_range_check': (C.c_int, [_VP, _VP]),
    'hqt_param_count': (C.c_int64, [_VP, C.c_int]),
    'hqt_workspace_bytes': (C.c_int64, [_VP]),
    'hqt_timing_enable': (C.c_int, [_VP, C.c_int]),
    'hqt_timing_reset': (C.c_int, [_VP]),
    'hqt_timing_get': (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'hqt_timing_slots': (C.c_int, [_VP]),
    'hqt_destroy': (C.c_int, [_VP]),
}

_lib: Optional[C.CDLL] = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libhqt.so for gfx950 in-tree with hipcc (cross-compiles without a GPU): one object per source, the stale
    ones in parallel, then one link.  "Stale" is decided by CONTENT, not by file times: every object carries a stamp = SHA-256 of its
    source, every header, the flags and `hipcc --version` (build/<name>.stamp), and libhqt.so one over the objects' stamps -- an object or
    a library shipped from another checkout, or left behind by an older source with a newer mtime, is rebuilt instead of loaded."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hdrs = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(os.path.dirname(HERE), 'include', 'hqt.h')])
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result']
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    ver = subprocess.run([hipcc, '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    common = hashlib.sha256()
    for f in hdrs:
        with open(f, 'rb') as fp:
            common.update(os.path.basename(f).encode() + b'\0' + fp.read())
    common.update(' '.join(flags).encode() + ver.encode())

    def read(path):
        try:
            with open(path) as fp:
                return fp.read().strip()
        except OSError:
            return ''

    def stamp_of(name):
        hh = common.copy()
        with open(os.path.join(CSRC, name), 'rb') as fp:
            hh.update(name.encode() + b'\0' + fp.read())
        return hh.hexdigest()

    def compile_one(name):
        src, obj = os.path.join(CSRC, name), os.path.join(objdir, name.replace('.hip', '.o'))
        stamp, want = obj[:-2] + '.stamp', stamp_of(name)
        if not force and os.path.exists(obj) and read(stamp) == want:
            return obj, want
        if os.path.exists(stamp):
            os.remove(stamp)
        cmd = [hipcc] + flags + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise HqtLibraryError(f'hipcc failed on {name}:\n' + proc.stdout)
        with open(stamp, 'w') as fp:
            fp.write(want + '\n')
        return obj, want

    with ThreadPoolExecutor(max_workers=min(len(SOURCES) + 1, max(1, (os.cpu_count() or 2) - 1))) as pool:
        audit = pool.submit(audit_hand_scheduled_loops, hipcc, flags, hdrs, force, verbose)
        built = list(pool.map(compile_one, SOURCES))
        audit.result()                               # raises HqtLibraryError: no library without a valid audit of the ISA it contains
    objs = [o for o, _ in built]
    lib_want = hashlib.sha256('\n'.join(st for _, st in built).encode()).hexdigest()
    lib_stamp = os.path.join(objdir, 'libhqt.stamp')
    if force or not os.path.exists(LIB_PATH) or read(lib_stamp) != lib_want:
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise HqtLibraryError('link failed:\n' + proc.stdout)
        with open(lib_stamp, 'w') as fp:
            fp.write(lib_want + '\n')
    return LIB_PATH


# kernels of split_stream_conv.hip whose main loop keeps inline-asm loads in flight across its back edge: EVERY instantiation the
# object holds is audited -- the symbols are taken from the generated ISA by name pattern, not from a list that a new template
# argument would silently miss
AUDITED_PATTERN = r'^\s*\.amdhsa_kernel\s+(\S*(?:conv3x3_split_ring16_kernel|conv2x2_split_up16_kernel|conv3x3_split_out16_kernel)\S*)'
AUDITED_MIN = 3                                   # at least the three kernels of round 4 must be there (a pattern that matches nothing is a broken audit)


def audit_hand_scheduled_loops(hipcc: str, flags: List[str], hdrs: List[str], force: bool = False, verbose: bool = False) -> None:
    """The SPLIT ring kernels (the default decode path) keep inline-asm loads in flight across their loop's back edge, behind the
    compiler's own waitcnt tracking.  That is only sound if hipcc leaves the loop one basic block and never touches a register with
    a load in flight -- checked on the ISA THIS toolchain generates from THESE sources with THESE flags (csrc/audit_ring.py).
    The verdict is cached under a hash of the source, every header, the flags and `hipcc --version`; a failed audit fails the build
    (HqtLibraryError), so neither build() of __graft_entry__ nor a direct rebuild can link an unaudited kernel."""
    import hashlib
    import sys
    import tempfile
    src = os.path.join(CSRC, 'split_stream_conv.hip')
    tool = os.path.join(CSRC, 'audit_ring.py')              # ships inside the package, next to the sources it audits
    if not os.path.exists(tool):
        raise HqtLibraryError(f'{tool} is missing: the ISA audit of the hand-scheduled SPLIT loops cannot run, refusing to build')
    stamp = os.path.join(CSRC, 'build', 'split_stream_conv.audit')
    hh = hashlib.sha256()
    for f in [src, tool] + sorted(hdrs):
        with open(f, 'rb') as fp:
            hh.update(f.encode() + b'\0' + fp.read())
    hh.update(' '.join(flags).encode())
    ver = subprocess.run([hipcc, '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    hh.update(ver.stdout.encode())
    hh.update(AUDITED_PATTERN.encode())
    key = hh.hexdigest()
    if not force and os.path.exists(stamp) and open(stamp).read().strip() == 'ok ' + key:
        return
    if os.path.exists(stamp):
        os.remove(stamp)
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'split_stream_conv.s')
        cmd = [hipcc] + [f for f in flags if f != '-fPIC'] + ['-S', '--cuda-device-only', '-o', asm, src]
        if verbose:
            print(' '.join(cmd), flush=True)
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise HqtLibraryError('hipcc -S failed on split_stream_conv.hip (ISA audit):\n' + proc.stdout)
        import re
        with open(asm) as fp:
            kernels = sorted(set(m.group(1) for m in re.finditer(AUDITED_PATTERN, fp.read(), re.M)))
        if len(kernels) < AUDITED_MIN:
            raise HqtLibraryError(f'ISA audit: found {kernels} in the ISA of split_stream_conv.hip, expected at least {AUDITED_MIN} ring kernels')
        for kernel in kernels:
            r = subprocess.run([sys.executable, tool, asm, kernel], capture_output=True, text=True)
            if r.returncode != 0:
                raise HqtLibraryError(f'ISA audit of {kernel} failed -- the hand-scheduled loop is not safe with this toolchain:\n{r.stdout}{r.stderr}')
    with open(stamp, 'w') as fp:
        fp.write('ok ' + key + '\n')


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it carries its own HIP runtime, and libhqt.so must bind to THAT copy.  Loaded the other way round (libhqt.so
    # pulling in /opt/rocm's libamdhip64 before torch is imported) the process ends up with the system runtime under torch and
    # torch.cuda reports 'No HIP GPUs are available'.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HqtLibraryError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hqtransformer_amd has no CPU fallback)')
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HqtLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HqtLibraryError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    if lib.hqt_abi_version() != ABI_VERSION:
        raise HqtLibraryError(f'ABI version {lib.hqt_abi_version()} != {ABI_VERSION}: rebuild libhqt.so')
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        raise HqtError(code, load().hqt_last_error().decode('utf-8', 'replace'))


def exported_symbols() -> List[str]:
    return list(SYMBOLS)
