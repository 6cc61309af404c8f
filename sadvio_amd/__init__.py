"""sadvio_amd — MI355X-native sliding-window bundle-adjustment backend for SaDVIO's optimizer boundary.

The product is the C-ABI shared library `csrc/libsadvio_ba.so` (include/sadvio_ba.h); this package holds
the HIP sources, the ctypes harness binding (`capi`) and the synthetic window generator (`synthetic`).
"""
__all__ = ["capi", "synthetic"]
