"""GPU tests of the sub-proof polynomials on a coset-sharded commitment (include/ligero_hip.h: lg_subproof_points /
lg_subproof_finish, lg_stage_digests_pack / _unpack).

The polynomials of prove_interleaved / prove_linear_constraints / prove_quadratic_constraints (src/ligero/mod.rs:658, 731-736,
845-848) are interpolated from values on the size-2k domain, which live in the coset planes s = 0 (mod 4): the rank holding
a plane computes the values at its slots alone.  Here `world` sharded contexts live in ONE process on the one GPU of the test
box and the two all-gathers are device copies between them, so that every C-ABI call of the multi-GPU path runs and the merged
result can be compared bit for bit with the unsharded entry points and with the oracle."""
import numpy as np
import pytest

from conftest import random_mont

pytestmark = pytest.mark.gpu


from sharded_inprocess import merge_points as _merge, sharded_commit as _sharded_commit


@pytest.mark.parametrize("rows,k,world", [(20, 128, 2), (24, 128, 8), (12, 64, 4), (8, 4096, 2), (8, 8192, 2), (4, 8192, 8), (21 * 4, 16, 2)])
def test_sharded_subproof_points_match_the_unsharded_polynomials(oracle, rows, k, world):
    import ligero_amd
    from ligero_amd import _ffi
    from ligero_amd.sharded import HipStageBackend
    pre = random_mont(901, rows * k).reshape(rows, k, 4)
    r_il = random_mont(902, rows).reshape(rows, 4)
    r_a = random_mont(903, rows * k).reshape(rows, k, 4)
    r_q = random_mont(904, rows // 4).reshape(rows // 4, 4)
    with ligero_amd.LigeroCommitter(rows=rows, k=k) as c:
        _, root = c.encode_commit(pre, want_coeffs=False)
        want_il = c.interleaved_row_mul(r_il)[0]
        want_lin = c.linear_constraint_poly(r_a)[0]
        want_q = c.quadratic_constraint_poly(r_q)[0]
        # the points API on an ordinary context: one call serves every plane
        for which, ch, want in ((_ffi.LG_SUB_INTERLEAVED, r_il, want_il), (_ffi.LG_SUB_LINEAR, r_a, want_lin), (_ffi.LG_SUB_QUADRATIC, r_q, want_q)):
            pts, mask = c.subproof_points(which, ch)
            assert mask == (0x11111111 if which != _ffi.LG_SUB_INTERLEAVED else 0x01010101) & ((1 << (8 if k <= 4096 else 8 * (k // 4096))) - 1)
            assert np.array_equal(c.subproof_finish(which, pts), want)
    backends = [HipStageBackend(rows, k, device=0, world=world, rank=r) for r in range(world)]
    try:
        roots = _sharded_commit(backends, pre)
        assert all(r == root for r in roots)
        np_ = backends[0].nplanes
        for which, ch, want in ((_ffi.LG_SUB_INTERLEAVED, r_il, want_il), (_ffi.LG_SUB_LINEAR, r_a, want_lin), (_ffi.LG_SUB_QUADRATIC, r_q, want_q)):
            parts = [be.c.subproof_points(which, ch) for be in backends]
            merged, owner = _merge(parts, np_)
            step = 8 if which == _ffi.LG_SUB_INTERLEAVED else 4
            assert sorted(owner) == list(range(0, np_, step))              # every plane of the domain was served exactly once
            for (pts, mask), be in zip(parts, backends):                   # ... by the context that holds it, zeros elsewhere
                lo, hi = be.c.planes()[1], be.c.planes()[1] + be.c.planes()[2]
                assert all(lo <= s < hi for s in range(np_) if mask & (1 << s))
                for j in range(2 * k):
                    if not mask & (1 << (4 * (j % (np_ // 4)))) or (which == _ffi.LG_SUB_INTERLEAVED and j % 2):
                        assert not pts[j].any()
            # any context finishes (no commitment needed for that), here the last rank's
            assert np.array_equal(backends[-1].c.subproof_finish(which, merged), want)
        # the unsharded entry points on a sharded commitment still refuse (they would read planes that are not there)
        if world > 1:
            with pytest.raises(ligero_amd.LigeroHipError) as e:
                backends[0].c.quadratic_constraint_poly(r_q)
            assert e.value.status == _ffi.LG_ERR_STATE
    finally:
        for be in backends:
            be.close()
    # and against the oracle's restatement of the three sums
    ref = oracle.encode_commit(pre, k, 8 * k)
    assert np.array_equal(want_il, oracle.dense_row_mul(pre, r_il))
    if 2 * k <= 4096:                                                       # (the oracle's polynomial products are quadratic-time)
        assert np.array_equal(want_lin, oracle.linear_constraint_poly(ref["coeffs"], r_a))
        assert np.array_equal(want_q, oracle.quadratic_constraint_poly(ref["coeffs"], r_q))


def test_linear_points_from_seed_need_the_matrix_only_where_a_plane_lives():
    """LG_SUB_LINEAR_FROM_SEED: ChaCha20 challenges and A.row_mul on the device of the ranks that hold a plane of the size-2k
    domain; a rank that holds none returns zeros without a constraint matrix"""
    import ligero_amd
    from ligero_amd import _ffi
    from ligero_amd.sharded import HipStageBackend
    rows, k, world = 16, 64, 8
    pre = random_mont(77, rows * k).reshape(rows, k, 4)
    # a small sparse "constraint matrix" over the 4 m k = rows * k columns: the identity plus a few off-diagonal entries
    ncols = rows * k
    ri = np.concatenate([np.arange(ncols), np.arange(0, ncols, 7)]).astype(np.uint64)
    ci = np.concatenate([np.arange(ncols), (np.arange(0, ncols, 7) * 5 + 3) % ncols]).astype(np.uint64)
    vals = random_mont(78, ri.shape[0]).reshape(-1, 4)
    seed = bytes(range(32))
    with ligero_amd.LigeroCommitter(rows=rows, k=k) as c:
        c.encode_commit(pre, want_coeffs=False)
        c.upload_constraint_matrix(ncols, ri, ci, vals)
        want = c.linear_constraint_poly_from_seeds(seed)[0]
    backends = [HipStageBackend(rows, k, device=0, world=world, rank=r) for r in range(world)]
    try:
        _sharded_commit(backends, pre)
        parts = []
        for r, be in enumerate(backends):
            if r in (0, 4):                                                 # planes 0 and 4 live on ranks 0 and 4
                with pytest.raises(ligero_amd.LigeroHipError):             # no matrix yet
                    be.c.subproof_points(_ffi.LG_SUB_LINEAR_FROM_SEED, seed)
                be.c.upload_constraint_matrix(ncols, ri, ci, vals)
            parts.append(be.c.subproof_points(_ffi.LG_SUB_LINEAR_FROM_SEED, seed))
        assert [m for _, m in parts] == [1, 0, 0, 0, 16, 0, 0, 0]
        merged, _ = _merge(parts, 8)
        assert np.array_equal(backends[3].c.subproof_finish(_ffi.LG_SUB_LINEAR_FROM_SEED, merged), want)
    finally:
        for be in backends:
            be.close()
