"""peps_amd: MI355X-native boundary-MPS contraction path of QuantumLiquids/PEPS behind a C ABI.

`peps_amd.capi` binds libpepsgpu.so (HIP, gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"
