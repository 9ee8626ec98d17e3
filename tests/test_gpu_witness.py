"""GPU tests of a1 on the device (include/ligero_hip.h lg_upload_gate_map / lg_encode_commit_from_witness): the commit from
the solution vector w alone -- X, Y, Z of preenc_u gathered on the GPU by the circuit's wiring (src/ligero/mod.rs:483-516) --
must give the commitment lg_encode_commit gives for the host-assembled matrix, which the oracle and the goldens pin."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, random_mont
from test_host_pipeline import rebuild_preenc_from_w

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def poseidon_inst():
    from ligero_amd import host_pipeline as hp
    return hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs")))


def _batch_witnesses(oracle, count):
    blob = open(os.path.join(GOLDEN, "poseidon_witness_batch64.bin"), "rb").read()
    out = []
    for i in range(count):
        ints = [int.from_bytes(blob[(i * 265 + j) * 32:(i * 265 + j + 1) * 32], "little") for j in range(1, 265)]
        out.append(oracle.to_mont(oracle.ints_to_limbs(ints)))
    return out


@pytest.mark.parametrize("batch", [1, 3, 64])
def test_poseidon_from_witness_equals_the_host_assembled_commit(oracle, vectors, poseidon_inst, batch):
    """Poseidon fixture (batch 1: the golden root of tests/golden/vectors.json) and the committed 64-witness batch: root,
    coefficient rows, resident preenc_u (through the interleaved test's row_mul) and openings equal the lg_encode_commit path"""
    import ligero_amd
    inst = poseidon_inst
    idx = list(range(1, 265))
    wit = _batch_witnesses(oracle, batch)
    pre = np.concatenate([inst.build_preenc_u(idx, v)[0] for v in wit])
    w = np.concatenate([inst.build_w(idx, v)[0] for v in wit])
    left, right, consts = inst.gate_map()
    with ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k, batch=batch) as ref, ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k, batch=batch) as c:
        coeffs_ref, root_ref = ref.encode_commit(pre)
        c.upload_gate_map(left, right, consts)
        for _ in range(2):                                                  # twice: the second commit overwrites a resident one
            coeffs, root = c.encode_commit_from_witness(w, want_coeffs=True)
            assert root == root_ref
            assert np.array_equal(coeffs, coeffs_ref)
        if batch == 1:
            assert root.hex() == vectors["poseidon"]["root"]
        r = random_mont(5, batch * inst.rows).reshape(batch * inst.rows, 4)
        assert np.array_equal(c.interleaved_row_mul(r), ref.interleaved_row_mul(r))      # the gathered X, Y, Z are in LG_BUF_PREENC
        assert np.array_equal(c.leaves(), ref.leaves())
        cols, sib, paths = c.open_columns([0, 9, 1023], proof=batch - 1)
        ecols, esib, epaths = ref.open_columns([0, 9, 1023], proof=batch - 1)
        assert np.array_equal(cols, ecols) and np.array_equal(sib, esib) and np.array_equal(paths, epaths)
        # an ordinary commit on the same context afterwards is unaffected
        assert c.encode_commit(pre, want_coeffs=False)[1] == root_ref


@pytest.mark.parametrize("m,k,batch,forward", [(6, 16, 2, False), (7, 64, 1, True), (5, 4096, 1, False), (3, 8192, 1, True), (40, 1024, 3, False)])
def test_synthetic_gate_maps_match_oracle(oracle, m, k, batch, forward):
    """hand-made wirings on random w: backward-only maps take the stepped upload (the transfer of w hides behind the encoding of
    complete rows), a map with FORWARD references falls back to uploading w first; odd m (odd block boundaries inside the column
    hash), constants, folded k = 8192, batches.  Expected commitment: the oracle on the numpy-rebuilt matrix."""
    import ligero_amd
    rng = np.random.default_rng(m * 1000 + k)
    mk = m * k
    npos = mk - 3                                                           # the tail of w is zero padding
    consts = random_mont(77, 5).reshape(5, 4)
    left = np.full(npos, 0xffffffff, dtype=np.uint32)
    right = left.copy()
    gates = np.sort(rng.choice(np.arange(1, npos), size=max(1, npos // 3), replace=False))
    for side in (left, right):
        src = np.array([rng.integers(0, npos if forward else p) for p in gates], dtype=np.uint32)
        use_const = rng.random(gates.shape[0]) < 0.2
        src[use_const] = 0x80000000 | rng.integers(0, 5, size=int(use_const.sum())).astype(np.uint32)
        side[gates] = src
    if forward:
        left[gates[0]] = npos - 1                                           # at least one operand after its gate
    w = np.zeros((batch, mk, 4), dtype=np.uint64)
    w[:, :npos] = random_mont(9, batch * npos).reshape(batch, npos, 4)
    pre = np.concatenate([rebuild_preenc_from_w(w[b], left, right, consts, m, k) for b in range(batch)])
    with ligero_amd.LigeroCommitter(rows=4 * m, k=k, batch=batch) as c:
        c.upload_gate_map(left, right, consts)
        coeffs, root = c.encode_commit_from_witness(w.reshape(batch * m, k, 4), want_coeffs=True)
        want = [oracle.encode_commit(pre[b * 4 * m:(b + 1) * 4 * m], k, 8 * k, want_u=False) for b in range(batch)]
        assert root == b"".join(x["root"] for x in want)
        assert np.array_equal(coeffs, np.concatenate([x["coeffs"] for x in want]))
        assert np.array_equal(c.leaves(), np.stack([x["leaves"] for x in want]))


@pytest.mark.parametrize("m,k,pad_rows,chunks", [(40, 1024, 13, 3), (24, 512, 0, 4), (9, 256, 8, 2), (12, 128, 3, 0)])
def test_commit_while_w_is_still_being_produced(oracle, monkeypatch, m, k, pad_rows, chunks):
    """lg_encode_commit_from_witness_progress: another thread writes w front to back and publishes how many leading positions are
    final; the commit ships rows as they become final (the padding rows behind the solution vector first, then steps over the rows
    it reaches, all four blocks per step) and must equal the commit from a finished w.  Until a position is published the buffer
    holds GARBAGE there: a row shipped too early changes the root.  Pipelined plan forced at a small size (LG_FORCE_CHUNKS); the
    last case is a single-chunk commit, which waits for all of w."""
    import ctypes
    import threading
    import time
    import ligero_amd
    from ligero_amd import _ffi
    if chunks:
        monkeypatch.setenv("LG_FORCE_CHUNKS", str(chunks))
    rng = np.random.default_rng(m + k)
    mk = m * k
    npos = mk - pad_rows * k - 5
    left = np.full(npos, 0xffffffff, dtype=np.uint32)
    right = left.copy()
    gates = np.sort(rng.choice(np.arange(1, npos), size=npos // 4, replace=False))
    left[gates] = np.array([rng.integers(0, p) for p in gates], dtype=np.uint32)
    right[gates] = np.array([rng.integers(0, p) for p in gates], dtype=np.uint32)
    consts = np.zeros((0, 4), dtype=np.uint64)
    final = np.zeros((mk, 4), dtype=np.uint64)
    final[:npos] = random_mont(5, npos).reshape(npos, 4)
    pre = rebuild_preenc_from_w(final, left, right, consts, m, k)
    want = oracle.encode_commit(pre, k, 8 * k, want_u=False)["root"]
    L = _ffi.lib()
    with ligero_amd.LigeroCommitter(rows=4 * m, k=k) as c:
        c.upload_gate_map(left, right, consts)
        for rep in range(3):
            buf = np.ascontiguousarray(final.copy())
            buf[:npos] = np.uint64(0x1111111111111111)          # not yet produced
            ready = np.zeros(1, dtype=np.uint64)
            root = (ctypes.c_uint8 * 32)()

            def producer():
                step = max(1, npos // 23)
                for a in range(0, npos, step):
                    b = min(npos, a + step)
                    buf[a:b] = final[a:b]
                    ready[0] = b                                 # (CPython: the store above is complete before this one)
                    time.sleep(0.002)
                ready[0] = mk
            t = threading.Thread(target=producer)
            t.start()
            st = L.lg_encode_commit_from_witness_progress(c._ctx, buf.ctypes.data_as(ctypes.c_void_p), ready.ctypes.data_as(ctypes.c_void_p), None,
                                                          ctypes.cast(root, ctypes.c_void_p))
            t.join()
            _ffi.check(st, "lg_encode_commit_from_witness_progress", c._ctx)
            assert bytes(root) == want, rep


def test_refusals(oracle):
    import ligero_amd
    from ligero_amd import _ffi
    with ligero_amd.LigeroCommitter(rows=8, k=16) as c:
        w = random_mont(1, 2 * 16).reshape(2, 16, 4)
        with pytest.raises(_ffi.LigeroHipError) as e:                       # no gate map yet
            c.encode_commit_from_witness(w)
        assert e.value.status == _ffi.LG_ERR_STATE
        none = np.full(4, 0xffffffff, dtype=np.uint32)
        bad = none.copy()
        bad[1] = 40                                                         # a position outside the solution vector
        with pytest.raises(_ffi.LigeroHipError):
            c.upload_gate_map(bad, bad, np.zeros((0, 4), dtype=np.uint64))
        half = none.copy()
        half[2] = 1                                                         # a left operand without a right one
        with pytest.raises(_ffi.LigeroHipError):
            c.upload_gate_map(half, none, np.zeros((0, 4), dtype=np.uint64))
        c.upload_gate_map(none, none, np.zeros((0, 4), dtype=np.uint64))   # no gates at all: X = Y = Z = 0
        pre = np.zeros((8, 16, 4), dtype=np.uint64)
        pre[6:] = w
        assert c.encode_commit_from_witness(w)[1] == oracle.encode_commit(pre, 16, 128, want_u=False)["root"]


def test_prover_paths_agree(oracle, poseidon_inst, monkeypatch):
    """HipLigero::prove uploads w alone by default; LG_PREENC_ON_HOST=1 keeps the host-assembled matrix: same proof"""
    from ligero_amd.prover import LigeroProver, proofs_equal
    vals = _batch_witnesses(oracle, 1)[0]
    idx = list(range(1, 265))
    with LigeroProver(poseidon_inst) as a:
        pa = a.prove(idx, vals)
        monkeypatch.setenv("LG_PREENC_ON_HOST", "1")
        pb = a.prove(idx, vals)
        monkeypatch.delenv("LG_PREENC_ON_HOST")
        pc = a.prove(idx, vals)
        assert proofs_equal(pa, pb) and proofs_equal(pa, pc) and a.verify(pa)
