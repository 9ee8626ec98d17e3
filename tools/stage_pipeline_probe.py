#!/usr/bin/env python3
"""lg_stage_evaluate_hash with and without the row-chunk pipeline (LG_FORCE_CHUNKS=1 = one chunk, no overlap), one rank, S20 shape:
    python tools/stage_pipeline_probe.py [s20|s22]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LARGE_SEED, WORKLOADS, synthetic_preenc  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s20"
rows, k, _ = WORKLOADS[wl]
pre = synthetic_preenc(LARGE_SEED, rows * k).reshape(rows, k, 4)
for force in ("1", None):
    if force:
        os.environ["LG_FORCE_CHUNKS"] = force
    else:
        os.environ.pop("LG_FORCE_CHUNKS", None)
    from ligero_amd.sharded import HipStageBackend
    be = HipStageBackend(rows, k, world=1, rank=0)
    be.stage_interpolate(pre, 0, rows)
    be.sync()
    planes = list(range(be.nplanes))
    for rep in range(3):
        t = time.perf_counter()
        be.stage_evaluate_hash(planes)
        be.sync()
        dt = (time.perf_counter() - t) * 1e3
    be.stage_merkle()
    print(f"{wl} chunks={'1 (serial)' if force else 'planned'}: evaluate+hash of {len(planes)} planes {dt:.2f} ms  root {be.root().hex()[:16]}", flush=True)
    be.close()
