"""sm_hpss_mtl_amd -- MI355X (gfx950) implementation of the SM_HPSS_MTL hot path.

STFT -> harmonic/percussive median filtering + soft masks -> mel / dB -> standardise -> patches ->
B3_MTL forward, as hand-written HIP behind the C ABI of include/smh.h (libsmh.so).  The Python layer
mirrors the reference's call surface (`sm_hpss_mtl_amd.lib.preprocessing`, `.lib.cython_impl.tools`,
`.lib.proposed_architectures`; top-level `lib/` re-exports them under the reference's module paths).
There is no CPU fallback: importing is cheap, computing requires the built library and a GPU.
"""
__version__ = "0.1.0"
