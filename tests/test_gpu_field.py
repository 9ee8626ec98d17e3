"""The reference's second element type: `LigeroCircuit<F>` is generic over `F: PrimeField`
(src/ligero/mod.rs:146) and src/ligero/tests.rs instantiates ark_bn254::Fr AND ark_bls12_377::Fq (tests.rs:23, 186-193: 377 bits,
6 x u64 limbs, 48-byte canonical serialization).  The hot path for a generic field goes through the portable kernels
(ligero_amd/csrc/generic_kernels.h):
  * over BN254 Fr they must reproduce the C oracle AND the tuned path bit for bit (a cross-check of the kernels themselves);
  * over BLS12-377 Fq they must reproduce the Python big-int model (oracle/model_field.py) -- coefficients, codeword, the
    Blake2s leaves over 48-byte elements, tree, root, openings."""
import os

import numpy as np
import pytest

from conftest import random_mont

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lg():
    import ligero_amd
    from ligero_amd import _ffi
    _ffi.lib()
    return ligero_amd


@pytest.mark.parametrize("rows,k,batch", [(1, 2, 1), (3, 4, 1), (5, 8, 2), (12, 64, 1), (7, 128, 3), (344, 128, 1), (3, 1024, 1), (2, 4096, 1)])
def test_generic_kernels_on_bn254_match_oracle_and_tuned_path(lg, oracle, rows, k, batch):
    from ligero_amd import _ffi
    n = 8 * k
    pre = random_mont(17 * rows + k, batch * rows * k).reshape(batch * rows, k, 4)
    with lg.LigeroCommitter(rows=rows, k=k, batch=batch, field=_ffi.LG_FIELD_BN254_FR_GENERIC) as g, lg.LigeroCommitter(rows=rows, k=k, batch=batch) as f:
        assert g.ew == 4
        gco, groot = g.encode_commit(pre)
        fco, froot = f.encode_commit(pre)
        assert groot == froot and np.array_equal(gco, fco)
        assert np.array_equal(g.leaves(), f.leaves()) and np.array_equal(g.nodes(), f.nodes())
        idx = sorted({0, 1, min(9, n - 1), n - 1})
        for b in range(batch):
            ref = oracle.encode_commit(pre[b * rows:(b + 1) * rows], k, n)
            assert np.array_equal(gco[b * rows:(b + 1) * rows], ref["coeffs"])
            assert np.array_equal(g.codeword_rows(proof=b), ref["u"])
            assert groot[32 * b:32 * b + 32] == ref["root"]
            cols, sib, paths = g.open_columns(idx, proof=b)
            ecols, esib, epaths = oracle.open_columns(ref["u"], ref["leaves"], ref["nodes"], idx)
            assert np.array_equal(cols, ecols) and np.array_equal(sib, esib) and np.array_equal(paths, epaths)
        # row operators (mod.rs:998-1012)
        assert np.array_equal(g.reed_solomon_interpolate(pre[:rows]), fco[:rows])
        assert np.array_equal(g.reed_solomon(pre[:1]), f.reed_solomon(pre[:1]))
        assert np.array_equal(g.reed_solomon_evaluate(fco[:1]), f.codeword_rows(0, 1))


def _fq_rows(fq, seed, rows, k):
    rng = np.random.default_rng(seed)
    ints = [[int.from_bytes(rng.bytes(48), "little") % fq.p for _ in range(k)] for _ in range(rows)]
    # field corners
    ints[0][0] = 0
    ints[-1][-1] = fq.p - 1
    if k > 1:
        ints[0][1] = 1
    return ints


@pytest.mark.parametrize("rows,k,batch", [(1, 2, 1), (2, 2, 1), (3, 4, 1), (4, 4, 2), (5, 16, 1), (6, 64, 2), (9, 32, 1), (2, 2048, 1)])
def test_bls12_377_fq_hot_path_matches_model(lg, rows, k, batch):
    from ligero_amd import _ffi
    from oracle import model_field as mf
    fq = mf.BLS12_377_FQ
    n = 8 * k
    ints = [_fq_rows(fq, 100 * b + rows + k, rows, k) for b in range(batch)]
    pre = np.concatenate([fq.to_mont_limbs([v for row in m for v in row]).reshape(rows, k, 6) for m in ints])
    with lg.LigeroCommitter(rows=rows, k=k, batch=batch, field=_ffi.LG_FIELD_BLS12_377_FQ) as c:
        assert c.ew == 6
        coeffs, roots = c.encode_commit(pre)
        leaves, nodes = c.leaves(), c.nodes()
        for b in range(batch):
            eco, eu, elv, enodes, eroot = fq.encode_commit(ints[b], k, n)
            assert fq.from_mont_limbs(coeffs[b * rows:(b + 1) * rows]) == [v for r in eco for v in r], "coefficients (mod.rs:521-526)"
            assert fq.from_mont_limbs(c.codeword_rows(proof=b)) == [v for r in eu for v in r], "codeword (mod.rs:528-533)"
            assert [x.tobytes() for x in leaves[b]] == elv, "Blake2s over LE64(rows) || 48-byte elements (mod.rs:536-542)"
            assert [x.tobytes() for x in nodes[b]] == enodes and roots[32 * b:32 * b + 32] == eroot
            # systematic code: U[i][8 q] = message[i][q]
            cw = c.codeword_rows(proof=b)
            assert np.array_equal(cw[:, ::8, :], pre[b * rows:(b + 1) * rows])
            idx = sorted({0, 3 % n, n - 1})
            cols, sib, paths = c.open_columns(idx, proof=b)
            from oracle import model
            for i, j in enumerate(idx):
                assert fq.from_mont_limbs(cols[i]) == [row[j] for row in eu]
                esib, epath = model.merkle_path(elv, enodes, j)
                assert sib[i].tobytes() == esib and [x.tobytes() for x in paths[i]] == epath
                assert model.merkle_verify(eroot, fq.col_hash([row[j] for row in eu]), j, esib, epath)
        # resident entry points and the verifier's row operator
        c.upload(pre)
        c.commit_resident()
        assert c.root() == roots
        assert np.array_equal(c.reed_solomon_interpolate(pre[:rows]), coeffs[:rows])
        assert np.array_equal(c.reed_solomon(pre[:1]), c.codeword_rows(0, 1))


def test_generic_context_limits_and_unsupported_calls(lg):
    from ligero_amd import _ffi
    with pytest.raises(lg.LigeroHipError) as e:
        lg.LigeroCommitter(rows=2, k=4096, field=_ffi.LG_FIELD_BLS12_377_FQ)       # one row must fit LDS: k <= 2048 at 48 bytes per element
    assert e.value.status == _ffi.LG_ERR_UNSUPPORTED
    with lg.LigeroCommitter(rows=4, k=8, field=_ffi.LG_FIELD_BLS12_377_FQ) as c:
        with pytest.raises(lg.LigeroHipError) as e:
            c.root()                                                               # nothing committed yet
        assert e.value.status == _ffi.LG_ERR_STATE
        with pytest.raises(lg.LigeroHipError) as e:
            c.quadratic_constraint_poly(np.zeros((1, 6), dtype=np.uint64))         # needs a commitment
        assert e.value.status == _ffi.LG_ERR_STATE
        with pytest.raises(lg.LigeroHipError) as e:
            c.profile(True)                                                         # tuned-path-only call
        assert e.value.status == _ffi.LG_ERR_UNSUPPORTED


@pytest.mark.parametrize("rows,k", [(12, 8), (16, 4), (344, 128), (8, 1024)])
def test_generic_sub_proof_polynomials_on_bn254_match_oracle(lg, oracle, rows, k):
    """the three sub-proof reductions of the portable path (mod.rs:658, 723-736, 842-848), instantiated for BN254, against the
    oracle -- the same kernels serve ark_bls12_377::Fq in the prove / verify mirror below"""
    from ligero_amd import _ffi
    m = rows // 4
    pre = random_mont(3 * rows + k, rows * k).reshape(rows, k, 4)
    r_int = random_mont(21, rows).reshape(rows, 4)
    r_a = random_mont(22, rows * k).reshape(rows, k, 4)
    r_q = random_mont(23, m).reshape(m, 4)
    with lg.LigeroCommitter(rows=rows, k=k, field=_ffi.LG_FIELD_BN254_FR_GENERIC) as c:
        coeffs, _ = c.encode_commit(pre)
        assert np.array_equal(c.interleaved_row_mul(r_int)[0], oracle.dense_row_mul(pre, r_int))
        assert np.array_equal(c.linear_constraint_poly(r_a)[0], oracle.linear_constraint_poly(coeffs, r_a))
        assert np.array_equal(c.quadratic_constraint_poly(r_q)[0], oracle.quadratic_constraint_poly(coeffs, r_q))


def test_prove_and_verify_bls12_377():
    """src/ligero/tests.rs:186-193 mirrored in C++ over the templated host classes and the generic-field device path
    (ligero_amd/host/example_bls12_377.cpp): y^2 = x^3 + 1 over Fq, (m, k) = (4, 4), a curve point is accepted, x + 1 is rejected"""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "ligero_amd", "host", "example_bls12_377")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "test_prove_and_verify_bls12_377: ok" in r.stdout
    assert "test_construction_bls12_377 (A matrix tables): ok" in r.stdout       # tests.rs:35-142, the induced P_x / P_y / P_z / P_add
    assert r.stdout.count("valid assignment accepted, x + 1 rejected") == 3       # G, 2 G, and G on the expression-made circuit
    assert "prove_with_labels(2G) accepted" in r.stdout
