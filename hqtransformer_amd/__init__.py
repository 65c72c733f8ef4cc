"""MI355X-native HQ-Transformer sampling path.

Import is cheap; the HIP library is loaded on first use by ``hqtransformer_amd._lib``.
"""
__version__ = '0.1.0'
