"""MI355X-native (gfx950) implementation of the Effi-MVS+ cost-volume hot path.

Host side mirrors the reference's ``models`` package (same class names, signatures, state-dict keys);
the arithmetic runs in hand-written HIP kernels behind a C ABI (``include/effi_mvs_hip.h``,
``effi_mvs_plus_amd/csrc``).  There is no CPU fallback: calling an op without the built library or
with CPU tensors raises.
"""
__version__ = "0.1.0"
