"""sxmc_amd -- MI355X-native implementation of sxmc's per-step NLL evaluation.

The product is libsxmc_hip.so (hand-written gfx950 kernels behind the C ABI of
include/sxmc_hip.h) plus the C++ mirror of the reference's pdfz / nll_kernels interface in
sxmc_amd/include/sxmc/.  The Python modules here are a thin ctypes layer with the reference's
names, used by the parity tests and the bench harness.
"""
from . import capi  # noqa: F401
from . import pdfz  # noqa: F401
from . import nll  # noqa: F401

__all__ = ["capi", "pdfz", "nll"]
