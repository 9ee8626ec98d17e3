#!/usr/bin/env python3
"""Random sequences of the library's commit entry points on ONE long-lived context (include/ligero_hip.h): the resident commit, the
commit from host buffers, the commit from `w`, the staged calls (all planes at once; rows then row ranges of the column hash, cut at
random rows), lg_commit_sharded and lg_commit_row_relay at world 1 -- each over one of two matrices chosen at random, each followed
at random by reads of the commitment (root, leaves, coefficient rows, openings, the quadratic-test polynomial, the interleaved
row product).  Every answer is compared with what a FRESH context gives for the same matrix through the plain path.  What it is
after: state carried from one entry-point family into the next -- which rows and planes the context believes it holds, parked hash
states, pending asynchronous hash / tree work, the canonical message copy.

    python tools/fuzz_api_sequences.py <seconds> [rows=24] [k=64] [seed=1] [batch=1] [cycles=1]        (rows = 4 m)

batch > 1: the three entry points that take a batch (resident, host buffers, from `w`), the reads per proof and for the whole batch.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ligero_amd  # noqa: E402
from ligero_amd import _ffi  # noqa: E402
from ligero_amd.sharded import HipRelayBackend, TorchComm  # noqa: E402

_vp = ctypes.c_void_p


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    cycles = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # > 1: that many runs of `seconds` each, every one on contexts of its own
    for cyc in range(cycles):                                  # (many TEARDOWNS after varied activity per process: tools/hang_hunt.sh)
        run_once(seconds, rows, k, seed + 1000 * cyc, batch)


def run_once(seconds, rows, k, seed, batch):
    assert rows % 4 == 0
    rng = np.random.default_rng(seed)
    n, m = 8 * k, rows // 4
    left = np.full(m * k, 0xffffffff, dtype=np.uint32)
    right = left.copy()
    gates = np.flatnonzero(rng.integers(3, size=m * k) == 0)
    left[gates] = rng.integers(m * k, size=gates.size)
    right[gates] = rng.integers(m * k, size=gates.size)

    def matrix():
        a = rng.integers(0, 2**62, size=(batch, 4, m * k, 4), dtype=np.uint64)
        a[..., 3] &= np.uint64((1 << 60) - 1)
        a[:, :3] = 0
        w = a[:, 3]
        a[:, 0][:, gates], a[:, 1][:, gates], a[:, 2][:, gates] = w[:, left[gates]], w[:, right[gates]], w[:, gates]
        return a.reshape(batch * rows, k, 4), np.ascontiguousarray(w).reshape(batch * m, k, 4)

    mats = [matrix(), matrix()]
    idx = sorted(set(int(x) for x in rng.integers(n, size=9)) | {0, n - 1})
    r_quad = rng.integers(0, 2**62, size=(batch * m, 4), dtype=np.uint64)
    r_int = rng.integers(0, 2**62, size=(batch * rows, 4), dtype=np.uint64)
    small = rows * k <= 1 << 16                              # the linear test's explicit challenge is rows * k elements per proof
    r_a = rng.integers(0, 2**62, size=(batch * rows, k, 4), dtype=np.uint64) if small else None
    ncols = rows * k
    a_ri = np.concatenate([np.arange(ncols), np.arange(0, ncols, 7)]).astype(np.uint64)
    a_ci = np.concatenate([np.arange(ncols), (np.arange(0, ncols, 7) * 5 + 3) % ncols]).astype(np.uint64)
    a_vals = rng.integers(0, 2**62, size=(a_ri.shape[0], 4), dtype=np.uint64)
    seeds = bytes(int(x) for x in rng.integers(256, size=32 * batch))
    bidx = np.stack([np.array(idx, dtype=np.uint32)[rng.permutation(len(idx))] for _ in range(batch)])      # per-proof order
    want = []
    for pre, _ in mats:                                       # the plain path on a fresh context
        with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as c:
            coeffs, root = c.encode_commit(pre)
            c.upload_constraint_matrix(ncols, a_ri, a_ci, a_vals)
            want.append({"lin": c.linear_constraint_poly(r_a) if small and 4 <= k <= 8192 else None,
                         "seed": c.linear_constraint_poly_from_seeds(seeds) if 4 <= k <= 8192 else None,
                         "root": root, "coeffs": coeffs, "leaves": c.leaves().copy(), "open": c.open_columns(idx, proof=batch - 1),
                         "openb": c.open_columns_batch(bidx),
                         "quad": c.quadratic_constraint_poly(r_quad) if 4 <= k <= 8192 else None, "int": c.interleaved_row_mul(r_int)})
    trace = open(os.environ["LG_FUZZ_TRACE"], "w") if os.environ.get("LG_FUZZ_TRACE") else None
    L = _ffi.lib()
    be = HipRelayBackend(rows, k) if batch == 1 else None
    c = be.c if be is not None else ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch)
    comm = TorchComm(None)
    c.upload_gate_map(left, right, np.zeros((0, 4), dtype=np.uint64))
    c.upload_constraint_matrix(ncols, a_ri, a_ci, a_vals)
    all_pos = rng.permutation(m * k).astype(np.uint32)
    c.upload_trace_program({"op": np.zeros(m * k, dtype=np.uint8), "left": left * 0 + 0xffffffff, "right": left * 0 + 0xffffffff,
                            "order": np.zeros(0, dtype=np.uint32), "level_off": np.zeros(1, dtype=np.uint64), "outputs": np.zeros(0, dtype=np.uint32)})
    allp = 8 if k <= 4096 else 8 * (k // 4096)
    counts = {}

    def note(what):
        if trace is not None:
            trace.write(f"    {what}\n")
            trace.flush()

    def resident(i):
        note("upload")
        c.upload(mats[i][0])
        for _ in range(int(rng.integers(1, 4))):
            note("commit_resident")
            c.commit_resident()

    def host(i):
        note("encode_commit")
        c.encode_commit(mats[i][0], want_coeffs=bool(rng.integers(2)))

    def witness(i):
        note("encode_commit_from_witness")
        c.encode_commit_from_witness(mats[i][1])

    def inputs(i):                                            # the trace on the device: here every position is an input (the scatter,
        note("encode_commit_from_inputs")                      # the gathers and the resident-plan commit between the other entry points)
        c.encode_commit_from_inputs(all_pos, mats[i][1].reshape(batch, m * k, 4)[:, all_pos])

    def staged_all(i):
        be.stage_interpolate(mats[i][0], 0, rows)
        _ffi.check(L.lg_stage_evaluate_hash(c._ctx, (1 << allp) - 1), "lg_stage_evaluate_hash", c._ctx)
        be.stage_merkle()

    def staged_rows(i):
        be.stage_interpolate(mats[i][0], 0, rows)
        cuts = sorted(set(int(x) for x in rng.integers(1, rows, size=int(rng.integers(0, 4))))) + [rows]
        order = int(rng.integers(2))
        if order == 0:
            be.stage_evaluate_rows(0, rows)
        a = 0
        for b in cuts:
            if order == 1:
                be.stage_evaluate_rows(a, b - a)
            groups = int(rng.choice([1, 2, allp]))
            per = allp // groups
            for g in range(groups):
                be.stage_hash_rows(g * per, per, a, b - a, a, rows)
            a = b
        be.stage_merkle()

    def sharded(i):
        p = mats[i][0]
        _ffi.check(L.lg_commit_sharded(c._ctx, comm.ptr(), p.ctypes.data_as(_vp), int(rng.integers(1, 5))), "lg_commit_sharded", c._ctx)

    def relay(i):
        be.commit_native(comm, rows, "contiguous", mats[i][0], plane_groups=int(rng.choice([0, 1, 2])))

    ops = [resident, host, witness, inputs, staged_all, staged_rows, sharded, relay] if batch == 1 else [resident, host, witness, inputs]

    def check(i, what):
        wnt = want[i]
        reads = rng.integers(2, size=9)
        note(f"check {list(map(int, reads))}")
        assert c.root() == wnt["root"], (what, "root")
        note("root ok")
        if reads[0]:
            assert np.array_equal(c.leaves(), wnt["leaves"]), (what, "leaves")
        if reads[1]:
            assert np.array_equal(c.coeffs(), wnt["coeffs"]), (what, "coeffs")
        if reads[2]:
            got = c.open_columns(idx, proof=batch - 1)
            assert all(np.array_equal(a, b) for a, b in zip(got, wnt["open"])), (what, "openings")
        if reads[6]:
            got = c.open_columns_batch(bidx)
            assert all(np.array_equal(a, b) for a, b in zip(got, wnt["openb"])), (what, "batch openings")
        if reads[3] and wnt["quad"] is not None:              # (the size-2k extraction serves 4 <= k <= 8192)
            assert np.array_equal(c.quadratic_constraint_poly(r_quad), wnt["quad"]), (what, "quadratic polynomial")
        if reads[4]:
            assert np.array_equal(c.interleaved_row_mul(r_int), wnt["int"]), (what, "interleaved row product")
        if reads[7] and wnt["lin"] is not None:
            assert np.array_equal(c.linear_constraint_poly(r_a), wnt["lin"]), (what, "linear polynomial")
        if reads[8] and wnt["seed"] is not None:
            assert np.array_equal(c.linear_constraint_poly_from_seeds(seeds), wnt["seed"]), (what, "linear polynomial from seeds")
        note("reads ok")
        if reads[5]:
            c.commit_resident()                               # ... and the matrix is still the resident one
            assert c.root() == wnt["root"], (what, "resident commit afterwards")

    if trace is not None:
        import faulthandler
        faulthandler.dump_traceback_later(seconds + 30, exit=True)     # a hang ends with the Python stack of the blocking call
    t_end = time.time() + seconds
    t_mark = time.time()
    trail = []
    total = 0
    try:
        while time.time() < t_end:
            op = ops[int(rng.integers(len(ops)))]
            i = int(rng.integers(2))
            trail.append((op.__name__, i))
            if trace is not None:                             # LG_FUZZ_TRACE=<file>: the op about to run (what a hang leaves behind)
                trace.write(f"{total} {op.__name__} {i}\n")
                trace.flush()
            op(i)
            if rng.integers(4):                               # sometimes the next entry point follows without any read in between
                check(i, trail[-6:])
            counts[op.__name__] = counts.get(op.__name__, 0) + 1
            total += 1
            if time.time() - t_mark > 30:
                t_mark = time.time()
                print(f"  ... {total} commits", flush=True)
    finally:
        (be if be is not None else c).close()
    print(f"api sequence fuzz ({batch} x {rows} x {k}, seed {seed}): {total} commits through {len(ops)} entry-point families on one context "
          f"({', '.join(f'{k_} {v}' for k_, v in sorted(counts.items()))}), every read equal to a fresh context's")


if __name__ == "__main__":
    main()
