"""Loader of the test-only proof-corruption hook (ligero_amd/lib/libligero_prover_testhooks.so, built from
ligero_amd/host/ligero_prover_testhooks.cpp).  The production prover library does not export it."""
import ctypes
import os

from conftest import ROOT

_lib = None


def tamper(proof, what: int, index: int = 0):
    """corrupt one item of `proof` (a ligero_amd.prover.Proof that owns its storage); raises RuntimeError on a borrowed view"""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(ROOT, "ligero_amd", "lib", "libligero_prover_testhooks.so"))
        _lib.lgp_proof_tamper.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
    rc = _lib.lgp_proof_tamper(proof._h, what, index)
    if rc != 0:
        raise RuntimeError(f"lgp_proof_tamper: status {rc}")
