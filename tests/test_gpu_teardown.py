"""lg_ctx_destroy_checked (include/ligero_hip.h): the streams of a context are drained under a deadline.  The ordinary case
releases everything; a context whose device work outlives the deadline is LEAKED and reported, naming the stream, instead of
blocking its caller inside hipStreamSynchronize for ever."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_LATE = r"""
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import ligero_amd
from ligero_amd import _ffi
rows, k = 10036, 4096                       # the S20 shape: about 21 ms of device work per commit
c = ligero_amd.LigeroCommitter(rows=rows, k=k)
L = _ffi.lib()
_ffi.check(L.lg_preenc_mark_filled(c._ctx), "lg_preenc_mark_filled", c._ctx)      # (whatever the buffer holds: only the time matters)
for _ in range(4):
    c.commit_resident()                     # queued, not waited for
t0 = time.time()
try:
    c.close()
    print("CLOSED")
except RuntimeError as e:
    print("LEAKED after %%.0f ms: %%s" %% ((time.time() - t0) * 1e3, e))
time.sleep(1.0)                             # the device finishes on its own; another context still works
with ligero_amd.LigeroCommitter(rows=8, k=16) as d:
    pre = np.zeros((8, 16, 4), dtype=np.uint64)
    d.encode_commit(pre, want_coeffs=False)
    print("NEXT CONTEXT OK")
""" % ROOT


def _run(deadline_ms):
    env = dict(os.environ, LG_TEARDOWN_TIMEOUT_MS=str(deadline_ms), LG_TRACE_TEARDOWN="1")
    r = subprocess.run([sys.executable, "-c", _LATE], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    return r.returncode, r.stdout, r.stderr


def test_teardown_waits_for_queued_work_within_the_deadline():
    rc, out, err = _run(60000)
    assert rc == 0 and "CLOSED" in out and "NEXT CONTEXT OK" in out, (out, err[-600:])
    assert "hipStreamQuery(main)" in err and "] done" in err          # every step traced before it is made


def test_a_context_whose_work_outlives_the_deadline_is_leaked_and_named():
    rc, out, err = _run(1)
    assert rc == 0 and "LEAKED" in out and "still holds unfinished work" in out and "NEXT CONTEXT OK" in out, (out, err[-600:])
    assert "context is leaked" in err


_IN_FLIGHT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
from ligero_amd import host_pipeline as hp
from ligero_amd.prover import LigeroBatchProver, LigeroProver
g = os.path.join(%r, "tests", "golden")
inst = hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(os.path.join(g, "poseidon.r1cs")))
w = hp.read_witness(os.path.join(g, "poseidon_witness.json"))
idx, vals = list(range(1, w.shape[0])), w[1:]
B = 96
allv = np.stack([vals] * B)
bp = LigeroBatchProver(inst, B, device_transcript=True)
bp.prove(idx, allv, copy=False)
bp.submit(idx, allv)
bp.submit(idx, allv)                         # two batches in flight: both arenas and both input blocks are being read and written
bp.close()                                   # as after an exception between submit() and collect(): must drain before it unregisters
print("CLOSED WITH TWO IN FLIGHT")
bp2 = LigeroBatchProver(inst, 8, device_transcript=True)
bp2.set_resident(True)
bp2.submit(idx, allv[:8])
bp2.close()
print("CLOSED RESIDENT IN FLIGHT")
with LigeroProver(inst) as p:
    assert p.verify(p.prove(idx, vals))
print("NEXT PROVER OK")
""" % (ROOT, ROOT)


def test_a_batch_prover_closed_with_batches_in_flight_drains_first():
    """round 5: HipLigeroBatch::release() collects what is in flight -- the encode stream, the prover's copy stream and the
    upload stream are still using the page-locked arenas and input blocks -- before it unregisters them and destroys the context; a
    process that closes a prover with two batches queued (what an exception between submit() and collect() leads to) must neither
    fault nor hang, and the device must be usable afterwards"""
    out = subprocess.run([sys.executable, "-c", _IN_FLIGHT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "CLOSED WITH TWO IN FLIGHT" in out.stdout and "CLOSED RESIDENT IN FLIGHT" in out.stdout and "NEXT PROVER OK" in out.stdout, out.stdout + out.stderr[-1000:]
