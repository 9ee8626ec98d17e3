"""GPU tests of f3 on the device (include/ligero_hip.h lg_upload_trace_program / lg_encode_commit_from_inputs): the evaluation
trace (src/arithmetic_circuit/mod.rs:325-358, called at src/ligero/mod.rs:476-478) run on the GPU level by level from the
prover's inputs alone must leave the bytes the host's evaluation leaves -- checked through everything that depends on them:
coefficient rows of all 4m rows (the interpolation is a bijection of the rows), leaves, root; the goldens pin the root."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, random_mont
from test_gpu_witness import _batch_witnesses

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def poseidon_inst():
    from ligero_amd import host_pipeline as hp
    return hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs")))


def _committers(inst, batch):
    import ligero_amd
    left, right, consts = inst.gate_map()
    ref = ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k, batch=batch)
    dev = ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k, batch=batch)
    for c in (ref, dev):
        c.upload_gate_map(left, right, consts)
    dev.upload_trace_program(inst.trace_program())
    return ref, dev


@pytest.mark.parametrize("batch", [1, 3, 64])
def test_poseidon_trace_on_the_device_equals_the_host_trace(oracle, vectors, poseidon_inst, batch):
    """the Poseidon fixture (batch 1: the golden root) and the committed 64-witness batch: 64 levels, 6 748 gates per proof"""
    inst = poseidon_inst
    idx = list(range(1, 265))
    wit = _batch_witnesses(oracle, batch)
    w = np.concatenate([inst.build_w(idx, v)[0] for v in wit])
    pos = inst.input_positions(idx)
    vals = np.stack(wit)
    ref, dev = _committers(inst, batch)
    with ref, dev:
        coeffs_ref, root_ref = ref.encode_commit_from_witness(w, want_coeffs=True)
        for _ in range(2):                                                  # twice: the second overwrites a resident commitment
            coeffs, root, ok = dev.encode_commit_from_inputs(pos, vals, want_coeffs=True)
            assert root == root_ref
            assert np.array_equal(coeffs, coeffs_ref)
            assert ok.all()
        if batch == 1:
            assert root.hex() == vectors["poseidon"]["root"]
        assert np.array_equal(dev.leaves(), ref.leaves())
        r = random_mont(5, batch * inst.rows).reshape(batch * inst.rows, 4)
        assert np.array_equal(dev.interleaved_row_mul(r), ref.interleaved_row_mul(r))      # LG_BUF_PREENC holds the same preenc_u
        # the same assignment in another order names the same variables
        perm = np.random.default_rng(1).permutation(len(idx))
        assert dev.encode_commit_from_inputs(pos[perm], vals[:, perm])[1] == root_ref
        # a wrong witness is a commitment to a trace whose outputs are not one -- for that proof only
        bad = vals.copy()
        bad[batch - 1, 7, 0] ^= 1
        _, root_bad, ok = dev.encode_commit_from_inputs(pos, bad)
        wb, ok_host = inst.build_w(idx, bad[batch - 1])
        assert not ok_host and not ok[batch - 1] and ok[:batch - 1].all()
        assert root_bad[:32 * (batch - 1)] == root_ref[:32 * (batch - 1)]
        w2 = w.copy().reshape(batch, -1, inst.k, 4)
        w2[batch - 1] = wb
        assert root_bad == ref.encode_commit_from_witness(w2.reshape(-1, inst.k, 4))[1]


def _random_circuit(seed, nvars, ngates, deep):
    from ligero_amd import host_pipeline as hp
    rng = np.random.default_rng(seed)
    c = hp.ArithmeticCircuit()
    consts = [c.constant(hp.fr_mont(int(v))) for v in (2, 5, 12345678901234567890123)]
    var = c.new_variables(nvars)
    live = list(var)                                       # nodes with a position
    used = set()
    gates = []
    for g in range(ngates):
        a = live[-1] if (deep and g % 3) else live[rng.integers(len(live))]      # deep: long dependency chains, narrow levels
        b = consts[rng.integers(len(consts))] if rng.random() < 0.2 else live[rng.integers(len(live))]
        if rng.random() < 0.5:
            a, b = b, a
        node = c.mul(a, b) if rng.random() < 0.5 else c.add(a, b)
        used.update((a, b))
        live.append(node)
        gates.append(node)
    loose = [n for n in gates + list(var) if n not in used]
    out = c.add_nodes(loose) if len(loose) > 1 else loose[0]
    return c, list(var), out


@pytest.mark.parametrize("seed,nvars,ngates,deep,batch", [(1, 5, 40, False, 1), (2, 12, 700, False, 3), (3, 3, 300, True, 2), (4, 40, 5000, False, 1)])
def test_builder_made_circuits(seed, nvars, ngates, deep, batch):
    """random add / mul circuits with constants through the builder API: wide and shallow, and deep chains (hundreds of levels of
    a few gates); the device trace against the host's on random assignments"""
    from ligero_amd import host_pipeline as hp
    circ, var, out = _random_circuit(seed, nvars, ngates, deep)
    inst = hp.LigeroInstance(circ, outputs=[out])
    prog = inst.trace_program()
    assert len(prog["level_off"]) - 1 >= (40 if deep else 2)
    vals = random_mont(seed + 10, batch * nvars).reshape(batch, nvars, 4)
    built = [inst.build_w(var, vals[b]) for b in range(batch)]
    w = np.concatenate([x[0] for x in built])
    ref, dev = _committers(inst, batch)
    with ref, dev:
        coeffs_ref, root_ref = ref.encode_commit_from_witness(w, want_coeffs=True)
        coeffs, root, ok = dev.encode_commit_from_inputs(inst.input_positions(var), vals, want_coeffs=True)
        assert np.array_equal(coeffs, coeffs_ref) and root == root_ref
        assert list(ok) == [x[1] for x in built]


def test_expression_made_circuit_and_refusals(poseidon_inst):
    import ligero_amd
    from ligero_amd import host_pipeline as hp
    x, y = hp.Expression.variable("x"), hp.Expression.variable("y")
    e = (x * x + 3) * (x + y) - y.pow(5) * 7 + x
    circ = e.to_arithmetic_circuit()
    inst = hp.LigeroInstance(circ, outputs=[circ.last()])
    labels = ["x", "y"]
    nodes = [circ.get_variable(l) for l in labels]
    vals = random_mont(3, 2).reshape(1, 2, 4)
    w, ok_host = inst.build_w(nodes, vals[0])
    ref, dev = _committers(inst, 1)
    with ref, dev:
        _, root, ok = dev.encode_commit_from_inputs(inst.input_positions(nodes), vals)
        assert root == ref.encode_commit_from_witness(w)[1] and bool(ok[0]) == ok_host
        pos = inst.input_positions(nodes)
        # every variable, once, and nothing else
        with pytest.raises(ligero_amd.LigeroHipError, match="Uninitialised variable"):
            dev.encode_commit_from_inputs(pos[:1], vals[:, :1])
        with pytest.raises(ligero_amd.LigeroHipError, match="assigned twice"):
            dev.encode_commit_from_inputs(np.array([pos[0], pos[0]], dtype=np.uint32), vals)
        with pytest.raises(ligero_amd.LigeroHipError, match="non-variable"):
            dev.encode_commit_from_inputs(np.array([pos[0], 0], dtype=np.uint32), vals)
        # the context still works
        assert dev.encode_commit_from_inputs(pos, vals)[1] == root
        # a program that is not a schedule is refused at upload: an operand in the same level as its gate
        prog = inst.trace_program()
        lo = prog["level_off"].copy()
        if len(lo) > 2:
            merged = dict(prog)
            merged["level_off"] = np.concatenate([lo[:1], lo[2:]])             # levels 1 and 2 fused
            with pytest.raises(ligero_amd.LigeroHipError):
                dev.upload_trace_program(merged)
        twice = dict(prog)
        twice["order"] = prog["order"].copy()
        twice["order"][0] = twice["order"][-1]                                  # a gate listed twice, another never
        with pytest.raises(ligero_amd.LigeroHipError):
            dev.upload_trace_program(twice)
        # a refused program leaves the loaded one in place
        assert dev.encode_commit_from_inputs(pos, vals)[1] == root
    # without the gate map there are no constants to share
    with ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k) as c:
        with pytest.raises(ligero_amd.LigeroHipError):
            c.upload_trace_program(prog)


def test_provers_make_the_same_proofs_with_the_trace_on_the_device(oracle, model, poseidon_inst, monkeypatch):
    """the single prover and the throughput-mode prover with the evaluation trace forced onto the device (LG_DEVICE_TRACE=1; a lone
    Poseidon proof would keep it on the host by the cost estimate) and forced off: the same proofs field for field, accepted; an
    assignment that names a variable twice (legal for the reference: the last value wins) and one that leaves a variable out (the
    reference's panic) behave as on the host"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroBatchProver, LigeroProver, proofs_equal
    inst = poseidon_inst
    idx = list(range(1, 265))
    wit = np.stack(_batch_witnesses(oracle, 5))
    monkeypatch.setenv("LG_DEVICE_TRACE", "0")
    with LigeroProver(inst) as host:
        ref = [host.prove(idx, wit[b]) for b in range(5)]
        monkeypatch.setenv("LG_DEVICE_TRACE", "1")
        with LigeroProver(inst) as dev:
            for b in range(5):
                p = dev.prove(idx, wit[b])
                assert proofs_equal(ref[b], p) and host.verify(p)
            # a variable named twice: the last value wins (the host evaluates this one)
            twice_idx = idx + [idx[3]]
            twice_val = np.concatenate([wit[0], wit[1][3:4]])
            fixed = wit[0].copy()
            fixed[3] = wit[1][3]
            assert proofs_equal(dev.prove(twice_idx, twice_val), host.prove(idx, fixed))
            # a variable left out: the reference's panic, in the reference's words
            with pytest.raises(Exception, match="Uninitialised variable"):
                dev.prove(idx[:-1], wit[0][:-1])
            assert proofs_equal(dev.prove(idx, wit[2]), ref[2])                   # ... and the prover still works
        with LigeroBatchProver(inst, 5, device_transcript=True) as bp:
            got = bp.prove(idx, wit)
            for b in range(5):
                assert proofs_equal(ref[b], got[b]), b
            bad = wit.copy()
            bad[4][10] = bad[4][11]
            views = bp.prove(idx, bad, copy=False)
            assert proofs_equal(views[0], ref[0]) and not host.verify(views[4]) and proofs_equal(views[4], host.prove(idx, bad[4]))
        monkeypatch.setenv("LG_DEVICE_TRACE", "0")
        with LigeroBatchProver(inst, 5, device_transcript=True) as bp:
            assert all(proofs_equal(ref[b], p) for b, p in enumerate(bp.prove(idx, wit)))


def test_tracer_rows_equal_the_host_assembled_rows(oracle, poseidon_inst):
    """lg_tracer_create / lg_tracer_rows (what a rank of a sharded proof uses): any row ranges of [X; Y; Z; W] from the assignment,
    in the order asked for, equal the rows of build_preenc_u; a second assignment on the same tracer; the refusals of the commit
    from inputs"""
    import ctypes
    from ligero_amd import _ffi
    L = _ffi.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    inst = poseidon_inst
    prog = inst.trace_program()

    class Desc(ctypes.Structure):
        _fields_ = [("m", ctypes.c_uint64), ("k", ctypes.c_uint32), ("npos", ctypes.c_uint64), ("op", ctypes.c_void_p), ("left", ctypes.c_void_p),
                    ("right", ctypes.c_void_p), ("constants", ctypes.c_void_p), ("nconst", ctypes.c_uint32), ("order", ctypes.c_void_p),
                    ("ngates", ctypes.c_uint64), ("level_off", ctypes.c_void_p), ("nlevels", ctypes.c_uint32), ("outputs", ctypes.c_void_p),
                    ("nout", ctypes.c_uint32)]
    keep = {k_: np.ascontiguousarray(v) for k_, v in prog.items() if isinstance(v, np.ndarray)}
    d = Desc(inst.m, inst.k, len(keep["op"]), keep["op"].ctypes.data, keep["left"].ctypes.data, keep["right"].ctypes.data, keep["constants"].ctypes.data,
             len(keep["constants"]), keep["order"].ctypes.data, len(keep["order"]), keep["level_off"].ctypes.data, len(keep["level_off"]) - 1,
             keep["outputs"].ctypes.data, len(keep["outputs"]))
    tr = ctypes.c_void_p()
    assert L.lg_tracer_create(ctypes.byref(tr), 0, ctypes.byref(d)) == 0, L.lg_tracer_last_error(None)
    try:
        idx = list(range(1, 265))
        pos = inst.input_positions(idx)
        m = inst.m
        for b, wit in enumerate(_batch_witnesses(oracle, 2)):
            pre, ok_host = inst.build_preenc_u(idx, wit)
            ranges = np.array([[3 * m, m], [0, 5], [m + 3, 7], [2 * m - 1, 2], [4 * m - 1, 1]], dtype=np.uint64)     # W whole, then bits of X, Y, Y|Z, W
            dev = ctypes.c_void_p()
            ok = ctypes.c_uint32(7)
            vals = np.ascontiguousarray(wit)
            st = L.lg_tracer_rows(tr, pos.ctypes.data, vals.ctypes.data, len(idx), ranges.ctypes.data, len(ranges), ctypes.byref(dev), ctypes.byref(ok))
            assert st == 0, L.lg_tracer_last_error(tr)
            total = int(ranges[:, 1].sum())
            got = np.empty((total, inst.k, 4), dtype=np.uint64)
            assert hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), dev, ctypes.c_size_t(got.nbytes), 2) == 0
            want = np.concatenate([pre[int(a):int(a + n)] for a, n in ranges])
            assert np.array_equal(got, want), b
            assert ok.value == int(ok_host) == 1
        # the same refusals as the commit from the inputs; the tracer stays usable
        dev = ctypes.c_void_p()
        assert L.lg_tracer_rows(tr, pos.ctypes.data, vals.ctypes.data, len(idx) - 1, ranges.ctypes.data, 1, ctypes.byref(dev), None) == _ffi.LG_ERR_BAD_ARG
        assert b"Uninitialised variable" in L.lg_tracer_last_error(tr)
        outside = np.array([[4 * m, 1]], dtype=np.uint64)
        assert L.lg_tracer_rows(tr, pos.ctypes.data, vals.ctypes.data, len(idx), outside.ctypes.data, 1, ctypes.byref(dev), None) == _ffi.LG_ERR_BAD_ARG
        assert L.lg_tracer_rows(tr, pos.ctypes.data, vals.ctypes.data, len(idx), ranges.ctypes.data, 0, ctypes.byref(dev), None) == 0      # a rank without rows
    finally:
        L.lg_tracer_destroy(tr)


def test_smallest_circuit():
    """two gates (the reference refuses an output that is not a gate): the scatter, two one-gate levels, the gathers and the commit"""
    from ligero_amd import host_pipeline as hp
    c = hp.ArithmeticCircuit()
    var = c.new_variables(3)
    inst = hp.LigeroInstance(c, outputs=[c.add(c.mul(var[0], var[1]), var[2])])
    prog = inst.trace_program()
    assert len(prog["order"]) == 2 and len(prog["level_off"]) - 1 == 2
    vals = random_mont(31, 3).reshape(1, 3, 4)
    w, ok_host = inst.build_w(var, vals[0])
    ref, dev = _committers(inst, 1)
    with ref, dev:
        _, root, ok = dev.encode_commit_from_inputs(inst.input_positions(var), vals)
        assert root == ref.encode_commit_from_witness(w)[1] and bool(ok[0]) == ok_host


def test_an_empty_assignment_on_a_fresh_program_is_refused(poseidon_inst):
    """round 5 (ADVICE): a freshly uploaded program remembers no assignment; an EMPTY one must not pass as "the same as last time" --
    the header promises LG_ERR_BAD_ARG ("Uninitialised variable", mod.rs:477) for a circuit that has variables, through the commitment
    entry point and through a rank's tracer, and the context must still work afterwards"""
    import ligero_amd
    from ligero_amd import _ffi
    inst = poseidon_inst
    left, right, consts = inst.gate_map()
    none_pos, none_vals = np.zeros(0, dtype=np.uint32), np.zeros((1, 0, 4), dtype=np.uint64)
    with ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k) as c:
        c.upload_gate_map(left, right, consts)
        c.upload_trace_program(inst.trace_program())
        with pytest.raises(ligero_amd.LigeroHipError) as e:
            c.encode_commit_from_inputs(none_pos, none_vals)
        assert e.value.status == _ffi.LG_ERR_BAD_ARG and "Uninitialised variable" in str(e.value)
        idx = list(range(1, 265))
        from test_gpu_witness import _batch_witnesses
        from oracle import binding as orc
        vals = np.stack(_batch_witnesses(orc, 1))
        _, root, ok = c.encode_commit_from_inputs(inst.input_positions(idx), vals)
        assert ok.all()
        with pytest.raises(ligero_amd.LigeroHipError):                       # and after a good assignment too
            c.encode_commit_from_inputs(none_pos, none_vals)
        assert c.encode_commit_from_inputs(inst.input_positions(idx), vals)[1] == root
