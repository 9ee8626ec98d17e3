"""Sanitizer runs of the HOST code, on the CPU build only (GPU sanitizers are not available on the pool):
  * AddressSanitizer + UBSan: mutation fuzz of the .r1cs / .wtns / witness.json readers (tools/fuzz_readers.py) -- the code that
    parses untrusted files (ligero_amd/host/circuit.hpp; the reference reads them through ark-circom, src/reader.rs)
  * ThreadSanitizer: the batch prover's host phases -- thread pool, per-proof transcripts, staging buffers
    (ligero_amd/host/prover.hpp HipLigeroBatch) -- over a race-detector stand-in for the device ABI (tests/sanitize/)
  * AddressSanitizer + UBSan over the CHECKER: oracle/ligero_oracle.c's commit, prover and verifier on the reference's small cases and
    random circuits, compared with the big-int model as tests/test_oracle_prover.py does -- a checker that reads out of bounds proves
    nothing."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT


def _gcc_file(name):
    return subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_asan_ubsan_fuzz_of_the_file_readers(tmp_path):
    asan = _gcc_file("libasan.so")
    if not os.path.isabs(asan):
        pytest.skip("libasan is not installed")
    out = os.path.join(ROOT, "build", "asan")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libligero_host.so")
    src = os.path.join(ROOT, "ligero_amd", "host", "ligero_host.cpp")
    deps = [src] + [os.path.join(ROOT, "ligero_amd", "host", h) for h in ("circuit.hpp", "expression.hpp", "field.hpp", "transcript.hpp")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fPIC", "-shared", "-o", lib, src])
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + " " + _gcc_file("libstdc++.so.6")
    env["ASAN_OPTIONS"] = "detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048:abort_on_error=0"
    env["LG_FUZZ_ITERS"] = "800"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_readers.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "fuzz done" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    ok, rejected = (int(x) for x in r.stdout.split("ok")[1].replace("rejected", "").split())
    assert ok > 0 and rejected > 0                      # both outcomes were exercised


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_tsan_of_the_batch_provers_host_phases(tmp_path):
    if not os.path.isabs(_gcc_file("libtsan.so")):
        pytest.skip("libtsan is not installed")
    exe = str(tmp_path / "tsan_host_phases")
    d = os.path.join(ROOT, "tests", "sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", exe,
                           os.path.join(d, "tsan_host_phases.cpp"), os.path.join(d, "stub_ligero_hip.cpp")])
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=1:exitcode=66"
    r = subprocess.run([exe, os.path.join(GOLDEN, "poseidon.r1cs"), os.path.join(GOLDEN, "poseidon_witness_batch64.bin")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "tsan harness done" in r.stdout, (r.returncode, r.stderr[-2000:])


_ORACLE_UNDER_ASAN = r"""
import sys
sys.path.insert(0, sys.argv[1])
from oracle import binding as orc, model_prover as MP
cases = [MP.lemniscate_circuit(), MP.multioutput_circuit()] + [MP.random_circuit(s, nvars=3, ngates=12 + s, one=o) for s, o in ((1, "first"), (2, "middle"), (3, "absent"))]
for threads in (1, 2):
    orc.lib().orc_prover_set_threads(threads)
    for circ, outs, va in cases:
        lc = MP.LigeroCircuit(circ, outs)
        st = orc.Statement(lc)
        fb = st.prove(va)
        assert st.verify(fb)
        proof = lc.prove_with_labels(va, MP.test_sponge()) if isinstance(va[0][0], str) else lc.prove(va, MP.test_sponge())
        assert fb == MP.proof_field_bytes(proof)
        bad = dict(fb); b = bytearray(bad["linear.columns"]); b[40] ^= 1; bad["linear.columns"] = bytes(b)
        assert not st.verify(bad)
print("oracle under asan done")
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_asan_ubsan_of_the_c_oracles_prover_and_verifier():
    asan = _gcc_file("libasan.so")
    if not os.path.isabs(asan):
        pytest.skip("libasan is not installed")
    out = os.path.join(ROOT, "build", "asan")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "liboracle_asan.so")
    src = os.path.join(ROOT, "oracle", "ligero_oracle.c")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-o", lib, src])
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + " " + _gcc_file("libstdc++.so.6")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"
    env["LIGERO_ORACLE_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", _ORACLE_UNDER_ASAN, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "oracle under asan done" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
