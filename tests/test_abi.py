"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol the header
declares, and fails loudly (no fallback) when there is no GPU.  No compute is launched."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ligero_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_match_binding():
    from ligero_amd import _ffi
    assert sorted(_ffi.SYMBOLS) == _declared_symbols()


def test_library_exports_every_declared_symbol():
    from ligero_amd import _ffi
    L = _ffi.lib()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (lg_[a-z0-9_]+)", out))
    for name in _declared_symbols():
        assert name in exported, name
        assert getattr(L, name) is not None
    assert L.lg_abi_version() == 4


def test_header_is_plain_c():
    """the boundary is a C ABI: the header must compile as C with no torch / C++ types"""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_status_strings():
    from ligero_amd import _ffi
    L = _ffi.lib()
    for code in range(0, -8, -1):
        assert L.lg_status_string(code)
    assert L.lg_status_string(0) == b"ok"


def test_argument_errors_need_no_gpu():
    from ligero_amd import _ffi
    L = _ffi.lib()
    ctx = ctypes.c_void_p()
    assert L.lg_ctx_create(None, 0, 16, 4, 32) == _ffi.LG_ERR_BAD_ARG
    assert L.lg_ctx_create(ctypes.byref(ctx), 0, 16, 6, 48) == _ffi.LG_ERR_BAD_DIMS      # k not a power of two
    assert L.lg_ctx_create(ctypes.byref(ctx), 0, 16, 4, 16) == _ffi.LG_ERR_BAD_DIMS      # n != 8k
    assert L.lg_ctx_create(ctypes.byref(ctx), 0, 0, 4, 32) == _ffi.LG_ERR_BAD_DIMS       # rows == 0
    assert L.lg_ctx_create(ctypes.byref(ctx), 0, 16, 1, 8) == _ffi.LG_ERR_BAD_DIMS       # k < 2
    assert L.lg_encode_commit(None, None, None, None) == _ffi.LG_ERR_BAD_ARG
    assert L.lg_open_columns(None, 0, None, 0, None, None, None) == _ffi.LG_ERR_BAD_ARG
    assert L.lg_sync(None) == _ffi.LG_ERR_BAD_ARG
    L.lg_ctx_destroy(None)


def test_no_fallback_without_gpu():
    """without a HIP device context creation must fail -- the product has no CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU boxes")
    import ligero_amd
    with pytest.raises(ligero_amd.LigeroHipError) as e:
        ligero_amd.LigeroCommitter(16, 4)
    assert e.value.status in (-3, -4)


def test_product_never_imports_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "ligero_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("#include", "import ", "from ")):
                        assert "oracle" not in s, (f, s)


def test_host_dimension_helpers(vectors):
    import ligero_amd
    assert ligero_amd.compute_dimensions(7274) == (86, 128)
    assert ligero_amd.compute_dimensions(15) == (4, 4)
    for k, t in vectors["calculate_t"].items():
        assert ligero_amd.reed_solomon_parameters(int(k), int(k), 128) == (8 * int(k), t)


def test_row_ownership_rules_of_the_library_equal_the_host_layer():
    """lg_shard_row_ranges / lg_relay_row_ranges (pure host functions of the C ABI: which rows a rank owns in the two
    multi-GPU modes) against ligero_amd.sharded's statement of the same rules, incl. BASELINE configs[3]'s 20 068 rows on 8 GPUs"""
    import ctypes
    import numpy as np
    from ligero_amd import _ffi
    from ligero_amd.sharded import relay_row_ranges, shard_row_ranges
    L = _ffi.lib()
    vp = ctypes.c_void_p
    for rows, world in [(20068, 8), (10036, 8), (344, 4), (7, 2), (3, 4), (1, 8), (64, 1)]:
        for pieces in (1, 2, 3, 4, 8, 50):
            for rank in range(world):
                out, n = np.zeros(16, dtype=np.uint32), ctypes.c_uint32(0)
                assert L.lg_shard_row_ranges(rows, world, rank, pieces, out.ctypes.data_as(vp), ctypes.cast(ctypes.byref(n), vp)) == 0
                assert [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value)] == shard_row_ranges(rows, world, rank, pieces)
        for layout, code in (("contiguous", _ffi.LG_RELAY_CONTIGUOUS), ("blocks", _ffi.LG_RELAY_BLOCKS), ("round_robin:2", _ffi.LG_RELAY_ROUND_ROBIN_BASE + 2),
                             ("round_robin:4", _ffi.LG_RELAY_ROUND_ROBIN_BASE + 4), ("round_robin:8", _ffi.LG_RELAY_ROUND_ROBIN_BASE + 8)):
            if layout == "blocks" and rows % 4:
                continue
            for rank in range(world):
                out, n = np.zeros(16, dtype=np.uint64), ctypes.c_uint32(0)
                assert L.lg_relay_row_ranges(rows, world, rank, code, out.ctypes.data_as(vp), ctypes.cast(ctypes.byref(n), vp)) == 0
                assert [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value)] == relay_row_ranges(rows, world, rank, layout)
    bad = np.zeros(8, dtype=np.uint64)
    assert L.lg_relay_row_ranges(10, 2, 0, _ffi.LG_RELAY_BLOCKS, bad.ctypes.data_as(vp), ctypes.cast(ctypes.byref(ctypes.c_uint32(0)), vp)) == _ffi.LG_ERR_BAD_ARG
    for code in (_ffi.LG_RELAY_ROUND_ROBIN_BASE, _ffi.LG_RELAY_ROUND_ROBIN_BASE + 9, 7):      # no such layout
        assert L.lg_relay_row_ranges(10, 2, 0, code, bad.ctypes.data_as(vp), ctypes.cast(ctypes.byref(ctypes.c_uint32(0)), vp)) == _ffi.LG_ERR_BAD_ARG
