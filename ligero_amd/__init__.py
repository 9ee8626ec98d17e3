"""ligero_amd: MI355X-native (gfx950) Ligero encode-and-commit hot path.

Host-side mirror of the pieces of NP-Eng/ligero's `LigeroCircuit` that sit on the path
src/ligero/mod.rs:521-551 and 935-955, driving hand-written HIP kernels through the C ABI
declared in include/ligero_hip.h.  See DESIGN.md.
"""
from ._ffi import LigeroHipError, LIB_PATH  # noqa: F401
from .ligero import LigeroCommitter, compute_dimensions, reed_solomon_parameters, calculate_t  # noqa: F401

__all__ = ["LigeroCommitter", "LigeroHipError", "compute_dimensions", "reed_solomon_parameters", "calculate_t", "LIB_PATH"]
