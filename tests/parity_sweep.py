#!/usr/bin/env python3
"""Randomised differential sweep (GPU library vs the oracle) over shapes the fixed test cases do not name:
random rows / k / batch, random data incl. field corners, both commit entry points, openings, sub-proof polynomials, the commit from w
alone (random wirings), the row relay and the coset-sharded commit over several contexts.
    python tests/parity_sweep.py [seconds] [seed]
Collected by pytest through tests/test_gpu_parity_sweep.py (fixed seed, bounded time); as a script it sweeps for as long
as asked with a time-derived seed.  Lives under tests/ because it uses the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import ligero_amd
from oracle import binding as oracle            # the checker
from conftest import random_mont

def sweep(budget: float, seed: int) -> int:
    """random cases until `budget` seconds have passed; returns how many ran (every one asserted bit-exact)"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    t_mark = time.time()
    n_cases = 0
    P_LIMBS = np.array([0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029], dtype=np.uint64)
    while time.time() < t_end:
        logk = int(rng.integers(1, 15))       # 13, 14: folded transforms (k = 8192, 16384)
        k = 1 << logk
        max_rows = max(4, min(400, (1 << 21) // (8 * k)))
        rows = int(rng.integers(1, max_rows // 4 + 1)) * 4
        batch = int(rng.choice([1, 1, 2, 3, 5]))
        if batch * rows * k * 8 > (1 << 22):
            batch = 1
        seed = int(rng.integers(1 << 30))
        pre = random_mont(seed, batch * rows * k).reshape(batch * rows, k, 4)
        # a quarter of the cases are matrices with the STRUCTURE of preenc_u: a random wiring on random w (a1 on the device, round 3)
        wiring = None
        if rng.integers(4) == 0:
            from test_host_pipeline import rebuild_preenc_from_w
            m, mk = rows // 4, (rows // 4) * k
            npos = max(2, mk - int(rng.integers(0, min(mk - 1, 5))))
            left = np.full(npos, 0xffffffff, dtype=np.uint32)
            right = left.copy()
            gates = np.sort(rng.choice(np.arange(1, npos), size=max(1, (npos - 1) // 3), replace=False))
            forward = bool(rng.integers(4) == 0)
            consts = random_mont(seed + 11, 3).reshape(3, 4)
            for side in (left, right):
                src = (rng.random(gates.shape[0]) * (npos if forward else gates)).astype(np.uint32)
                use_const = rng.random(gates.shape[0]) < 0.15
                src[use_const] = 0x80000000 | rng.integers(0, 3, size=int(use_const.sum())).astype(np.uint32)
                side[gates] = src
            w = np.zeros((batch, mk, 4), dtype=np.uint64)
            w[:, :npos] = random_mont(seed + 12, batch * npos).reshape(batch, npos, 4)
            for _ in range(8):                       # corner values go into w (the matrix is a function of it)
                w[int(rng.integers(batch)), int(rng.integers(npos))] = [0, 0, 0, 0] if rng.integers(2) else P_LIMBS - np.array([1, 0, 0, 0], dtype=np.uint64)
            pre = np.concatenate([rebuild_preenc_from_w(w[b], left, right, consts, m, k) for b in range(batch)])
            wiring = (w.reshape(batch * m, k, 4), left, right, consts)
        # sprinkle corner values: 0, 1 (Montgomery one is in random_mont's range anyway), p - 1
        for _ in range(0 if wiring is not None else 8):
            i, j = int(rng.integers(batch * rows)), int(rng.integers(k))
            pre[i, j] = [0, 0, 0, 0] if rng.integers(2) else P_LIMBS - np.array([1, 0, 0, 0], dtype=np.uint64)
        force = int(rng.choice([0, 0, 2, 3]))
        if force:
            os.environ["LG_FORCE_CHUNKS"] = str(force)
        else:
            os.environ.pop("LG_FORCE_CHUNKS", None)
        # both column-hash kernels and both stream layouts of single-chunk commits (round 2), whatever the size
        os.environ["LG_HASH_QUAD_MAX_COLUMNS"] = str(rng.choice([0, 32768, 1 << 40]))
        os.environ["LG_ASYNC_HASH"] = str(int(rng.integers(2)))
        with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as c:
            # other commitments first: the U double buffer, the parked Blake2s states and the asynchronous tree must not leak
            for _ in range(int(rng.integers(3))):
                other = random_mont(seed + 7, batch * rows * k).reshape(batch * rows, k, 4)
                if rng.integers(2):
                    c.encode_commit(other, want_coeffs=False)
                else:
                    c.upload(other); c.commit_resident()
            if rng.integers(2):
                coeffs, roots = c.encode_commit(pre)
            else:
                c.upload(pre); c.commit_resident(); coeffs, roots = c.coeffs(), c.root()
            t = min(8 * k, 5)
            idx = np.sort(rng.choice(8 * k, size=t, replace=False)).astype(np.uint32)
            for b in range(batch):
                ref = oracle.encode_commit(pre[b * rows:(b + 1) * rows], k, 8 * k)
                assert np.array_equal(coeffs[b * rows:(b + 1) * rows], ref["coeffs"]), ("coeffs", rows, k, batch, b, seed)
                assert roots[32 * b:32 * b + 32] == ref["root"], ("root", rows, k, batch, b, seed, force)
                cols, sib, paths = c.open_columns(idx, proof=b)
                ecols, esib, epaths = oracle.open_columns(ref["u"], ref["leaves"], ref["nodes"], idx)
                assert np.array_equal(cols, ecols) and np.array_equal(sib, esib) and np.array_equal(paths, epaths), ("open", rows, k, batch, b, seed)
            if k <= 4096 and rng.integers(4) == 0:
                # the portable generic-field kernels instantiated for BN254 must agree with the tuned path on everything
                from ligero_amd import _ffi
                with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch, field=_ffi.LG_FIELD_BN254_FR_GENERIC) as g:
                    gco, groots = g.encode_commit(pre)
                    assert groots == roots and np.array_equal(gco, coeffs), ("generic", rows, k, batch, seed)
                    assert np.array_equal(g.leaves(), c.leaves()), ("generic leaves", rows, k, batch, seed)
                    gc, gs, gp = g.open_columns(idx, proof=batch - 1)
                    tc, ts, tp = c.open_columns(idx, proof=batch - 1)
                    assert np.array_equal(gc, tc) and np.array_equal(gs, ts) and np.array_equal(gp, tp), ("generic open", rows, k, batch, seed)
            if k <= 8192 and rows % 4 == 0:
                r = random_mont(seed + 1, batch * rows // 4).reshape(batch, rows // 4, 4)
                got = c.quadratic_constraint_poly(r)
                for b in range(batch):
                    want = oracle.quadratic_constraint_poly(coeffs[b * rows:(b + 1) * rows], r[b])
                    assert np.array_equal(got[b], want), ("quad", rows, k, batch, b, seed)
            if wiring is not None:
                # the same commitment from w alone: X, Y, Z gathered on the device, the upload in steps (lg_encode_commit_from_witness)
                with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as cw:
                    cw.upload_gate_map(wiring[1], wiring[2], wiring[3])
                    wco, wroots = cw.encode_commit_from_witness(wiring[0], want_coeffs=True)
                    assert wroots == roots and np.array_equal(wco, coeffs), ("from witness", rows, k, batch, seed)
                    assert np.array_equal(cw.leaves(), c.leaves()), ("from witness leaves", rows, k, batch, seed)
            if batch == 1 and rng.integers(3) == 0:
                # the same proof ROW-sharded over 2 .. 5 contexts with the Blake2s states of the columns handed on (row relay, round 3):
                # both layouts, ragged and empty shards, odd boundaries; root, and every rank's rows of the opened columns
                from ligero_amd.sharded import HipRelayBackend, relay_chain
                from sharded_inprocess import relay_commit
                world = int(rng.integers(2, 6))
                layout = "blocks" if rng.integers(2) else "contiguous"
                chain = relay_chain(rows, world, layout)
                local = [sum(n for _, n, o, _ in chain if o == r) for r in range(world)]
                rbes = [HipRelayBackend(local[r], k, device=0) for r in range(world)]
                try:
                    assert all(rt == roots[:32] for rt in relay_commit(rbes, chain, pre, rows)), ("relay root", rows, k, world, layout, seed)
                    tc, ts, tp = c.open_columns(idx)
                    merged = np.empty_like(tc)
                    for r, be in enumerate(rbes):
                        gc, gs, gp = be.open_columns(idx)
                        assert np.array_equal(gs, ts) and np.array_equal(gp, tp), ("relay paths", rows, k, world, layout, seed)
                        for pos, n, o, loc in chain:
                            if o == r:
                                merged[:, pos:pos + n] = gc[:, loc:loc + n]
                    assert np.array_equal(merged, tc), ("relay columns", rows, k, world, layout, seed)
                finally:
                    for be in rbes:
                        be.close()
            if batch == 1 and k <= 8192 and rng.integers(4) == 0:
                # the same proof coset-sharded over 2 / 4 / 8 contexts (one process, exchanges as device copies): root, owner-served
                # openings, and the three sub-proof polynomials from the plane owners' point values
                from ligero_amd import _ffi
                from ligero_amd.sharded import HipStageBackend
                from sharded_inprocess import merge_points, sharded_commit
                world = int(rng.choice([2, 4, 8]))
                r_il = random_mont(seed + 2, rows).reshape(rows, 4)
                r_a = random_mont(seed + 3, rows * k).reshape(rows, k, 4)
                r_q = random_mont(seed + 4, rows // 4).reshape(rows // 4, 4)
                wants = ((_ffi.LG_SUB_INTERLEAVED, r_il, c.interleaved_row_mul(r_il)[0]), (_ffi.LG_SUB_LINEAR, r_a, c.linear_constraint_poly(r_a)[0]),
                         (_ffi.LG_SUB_QUADRATIC, r_q, c.quadratic_constraint_poly(r_q)[0]))
                bes = [HipStageBackend(rows, k, device=0, world=world, rank=r) for r in range(world)]
                try:
                    assert all(rt == roots[:32] for rt in sharded_commit(bes, pre)), ("sharded root", rows, k, world, seed)
                    per = bes[0].nplanes // world
                    for j in idx:
                        owner = (int(j) % bes[0].nplanes) // per
                        gc, gs, gp = bes[owner].open_columns([int(j)])
                        tc, ts, tp = c.open_columns([int(j)])
                        assert np.array_equal(gc, tc) and np.array_equal(gs, ts) and np.array_equal(gp, tp), ("sharded open", rows, k, world, seed)
                    for which, ch, want in wants:
                        merged, _ = merge_points([be.c.subproof_points(which, ch) for be in bes], bes[0].nplanes)
                        assert np.array_equal(bes[int(rng.integers(world))].c.subproof_finish(which, merged), want), ("sharded sub-proof", which, rows, k, world, seed)
                finally:
                    for be in bes:
                        be.close()
        n_cases += 1
        if time.time() - t_mark > 30.0:       # a long run must keep writing: the GPU box takes 7 silent minutes for a hang
            t_mark = time.time()
            print(f"  ... {n_cases} cases, {t_end - t_mark:.0f} s left", flush=True)
    for var in ("LG_FORCE_CHUNKS", "LG_HASH_QUAD_MAX_COLUMNS", "LG_ASYNC_HASH"):
        os.environ.pop(var, None)
    return n_cases


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    print(f"parity sweep (seed {seed}): {sweep(budget, seed)} random cases, all bit-exact")
