"""BASELINE.json configs[2] from an ACTUAL R1CS: tools/gen_repeated_squaring_r1cs.py writes the synthetic
2^20-constraint repeated-squaring system as a circom .r1cs + .wtns, the C++ host pipeline compiles it the
way ArithmeticCircuit::from_constraint_system does (src/arithmetic_circuit/mod.rs:455-520), LigeroCircuit::new derives
(m, k, n, t) = (2509, 4096, 32 768, 156) (src/ligero/mod.rs:171-175, 275-294), the witness goes through the evaluation trace into
preenc_u (mod.rs:476-516: x / y / z non-zero at Mul nodes only, w dense, zero tail), the GPU commits to it -- root against the
oracle's streamed restatement on the same matrix -- and prove() / verify() run end to end on the device prover."""
import importlib.util
import os
import time

import numpy as np
import pytest

from conftest import ROOT
from prover_hooks import tamper

pytestmark = pytest.mark.gpu


def _gen():
    spec = importlib.util.spec_from_file_location("gen_rs", os.path.join(ROOT, "tools", "gen_repeated_squaring_r1cs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_s20_from_r1cs_commit_prove_verify(oracle, tmp_path):
    import ligero_amd
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver
    gen = _gen()
    r1cs, wtns = str(tmp_path / "rs20.r1cs"), str(tmp_path / "rs20.wtns")
    gen.write_r1cs(r1cs, 20)
    wit = gen.witness(20, 1)
    gen.write_wtns(wtns, wit)
    t0 = time.time()
    circ = hp.ArithmeticCircuit.from_r1cs(r1cs)
    assert circ.num_nodes() == 5 * (1 << 20) + 3 == 5242883 and len(circ.outputs) == 1 << 20
    inst = hp.LigeroInstance(circ)
    assert (inst.m, inst.k, inst.n, inst.t) == (2509, 4096, 32768, 156)
    assert inst.num_constants == 2
    t_setup = time.time() - t0
    w = hp.read_witness(wtns)
    assert w.shape[0] == (1 << 20) + 2
    idx = np.arange(1, w.shape[0], dtype=np.uint64)
    pre, ok = inst.build_preenc_u(idx, w[1:])
    assert ok, "every output Az * Bz - Cz + 1 must evaluate to 1"
    # structure of preenc_u: the Mul-node positions of X / Y / Z, dense W, zero tail
    rows, k = 4 * inst.m, inst.k
    flat = pre.reshape(4, inst.m * k, 4)
    nz = [(flat[b] != 0).any(axis=1) for b in range(4)]
    sol_len = 1 + 5242883 - 2                                     # nodes kept in w: all but the non-leading constant
    # (one Add node per constraint holds a*b - c = 0, so 2^20 of the kept nodes are legitimately zero)
    assert int(nz[3].sum()) == sol_len - (1 << 20) and not nz[3][sol_len:].any() and nz[3][sol_len - 1]
    assert int(nz[0].sum()) == int(nz[1].sum()) == int(nz[2].sum()) == 2 * (1 << 20)   # two Mul gates per constraint
    with ligero_amd.LigeroCommitter(rows=rows, k=k) as c:
        _, root = c.encode_commit(pre, want_coeffs=False)
        want = oracle.encode_commit_streamed(pre, k, 8 * k, threads=min(16, os.cpu_count() or 1))
        assert root == want["root"]
        assert np.array_equal(c.leaves()[0], want["leaves"])
    with LigeroProver(inst) as prover:
        t1 = time.time()
        proof = prover.prove(idx, w[1:])
        t_first = time.time() - t1
        t1 = time.time()
        proof = prover.prove(idx, w[1:])
        t_prove = time.time() - t1
        info = proof.info()
        assert info["u_root"] == root
        assert info["opened_columns"] == 156 and info["column_len"] == rows and info["auth_path_len"] == 14
        # round 5: the WHOLE proof of BASELINE configs[2] is the oracle's proof, byte for byte in all ten fields (tests/golden/proofs_large.json:
        # oracle/model_prover.py's LigeroCircuit::new + oracle/ligero_oracle.c's orc_prove, made without the product)
        import json
        import proof_fp
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs_large.json")))["s20"]
        assert gold["dims"] == {"m": 2509, "k": 4096, "n": 32768, "t": 156} and gold["accepted"]
        fp = proof_fp.fingerprint(proof)
        assert proof_fp.same(fp, gold), proof_fp.diff(fp, gold)
        t1 = time.time()
        assert prover.verify(proof)
        t_verify = time.time() - t1
        print(f"s20 from r1cs: setup {t_setup:.1f} s, first prove {t_first:.2f} s, prove {t_prove:.2f} s, verify {t_verify:.1f} s")
        tamper(proof, 5, 12345)                                    # one element of an opened column of the linear test
        assert not prover.verify(proof)
        bad = w[1:].copy()
        bad[777] = bad[778]                                        # break one squaring
        assert not prover.verify(prover.prove(idx, bad))


@pytest.mark.parametrize("log_n", [10, 14])
def test_small_members_of_the_family_equal_the_oracles_proofs(tmp_path, log_n):
    """the same circuit family at 2^10 and 2^14 constraints (k = 128, 512): whole proofs = tests/golden/proofs_large.json, which the CPU
    suite regenerates from the oracle (tests/test_oracle_prover.py)"""
    import json
    import proof_fp
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver
    gen = _gen()
    r1cs, wtns = str(tmp_path / "rs.r1cs"), str(tmp_path / "rs.wtns")
    gen.write_r1cs(r1cs, log_n)
    gen.write_wtns(wtns, gen.witness(log_n, 1))
    inst = hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(r1cs))
    w = hp.read_witness(wtns)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs_large.json")))[f"s{log_n}"]
    assert (inst.m, inst.k, inst.n, inst.t) == tuple(gold["dims"][d] for d in "mknt")
    with LigeroProver(inst) as prover:
        proof = prover.prove(np.arange(1, w.shape[0], dtype=np.uint64), w[1:])
        fp = proof_fp.fingerprint(proof)
        assert proof_fp.same(fp, gold), proof_fp.diff(fp, gold)
        assert prover.verify(proof)


def test_s22_from_r1cs_prove_verify(tmp_path):
    """BASELINE configs[3] on ONE GPU, as a proof from the actual 2^22-constraint R1CS: (m, k, n, t) = (5017, 8192, 65 536, 156) as
    SURVEY 8d derives; U is 42 GB, the constraint matrix 186 M entries, the folded k = 8192 transforms and the 16-plane layout serve
    the commitment and all three sub-proofs; the verifier accepts, and rejects a proof whose opened column was altered.  (The
    commitment of this shape is pinned against the streamed oracle by test_gpu_parity.py on seeded data.)"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver
    gen = _gen()
    r1cs = str(tmp_path / "rs22.r1cs")
    gen.write_r1cs(r1cs, 22)
    wit = gen.witness(22, 1)
    t0 = time.time()
    circ = hp.ArithmeticCircuit.from_r1cs(r1cs)
    assert circ.num_nodes() == 5 * (1 << 22) + 3
    inst = hp.LigeroInstance(circ)
    assert (inst.m, inst.k, inst.n, inst.t) == (5017, 8192, 65536, 156)
    t_setup = time.time() - t0
    mask = (1 << 64) - 1
    vals = np.empty((len(wit) - 1, 4), dtype=np.uint64)
    for j, v in enumerate(wit[1:]):
        vm = (v << 256) % gen.P
        vals[j] = (vm & mask, (vm >> 64) & mask, (vm >> 128) & mask, vm >> 192)
    idx = np.arange(1, len(wit), dtype=np.uint64)
    with LigeroProver(inst) as prover:
        proof = prover.prove(idx, vals)
        t1 = time.time()
        proof = prover.prove(idx, vals)
        t_prove = time.time() - t1
        info = proof.info()
        assert info["opened_columns"] == 156 and info["column_len"] == 4 * 5017 and info["auth_path_len"] == 15
        assert info["linear_poly"] <= 2 * 8192 and info["quadratic_poly"] <= 2 * 8192 and info["preenc_u_lc"] == 8192
        t1 = time.time()
        assert prover.verify(proof)
        t_verify = time.time() - t1
        print(f"s22 from r1cs: setup {t_setup:.1f} s, prove {t_prove:.2f} s, verify {t_verify:.1f} s")
        tamper(proof, 5, 4321)
        assert not prover.verify(proof)
