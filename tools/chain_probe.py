#!/usr/bin/env python3
"""What one rank of an 8-GPU S22 commit would spend in the column hash, measured on ONE GPU (the launches are the ones the sharded
paths issue; only their concurrency with the other ranks is missing):
  row relay   : all 65 536 columns, rows / 8 rows        (lg_stage_hash_rows on a context of the rank's own rows)
  coset mode  : 2 of 16 planes = 8 192 columns, all rows (lg_stage_hash on a context that holds those planes)
and the same for 1, 2, 4 plane groups of the relay (is a launch over fewer columns faster?).  HIP-event timing through torch on the
library's stream."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ligero_amd import _ffi  # noqa: E402
from ligero_amd.sharded import HipRelayBackend, HipStageBackend  # noqa: E402

rows, k = 20068, 8192
G = 8
L = _ffi.lib()


def timed(fn, sync, reps=3):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


# ---- relay: a rank's own rows, all planes
local = rows // G
be = HipRelayBackend(local, k)
pre = bench.shard_rows_of_seeded_matrix(bench.LARGE_SEED, k, 0, local)
be.stage_interpolate(pre, 0, local)
be.stage_evaluate_rows(0, local)
be.sync()
t_enc = timed(lambda: (be.stage_interpolate(None, 0, local), be.stage_evaluate_rows(0, local)), be.sync)
print(f"relay rank: interpolate + evaluate {local} rows x 16 planes: {t_enc:.2f} ms")
for groups in (1, 2, 4, 16):
    per = be.nplanes // groups
    def go():
        for g in range(groups):
            be.stage_hash_rows(g * per, per, 0, local, 2 * (rows // G // 2) * 3, rows)      # a middle rank: resume and park
    print(f"relay rank: hash {local} rows, all 65536 columns in {groups:2d} plane group(s): {timed(go, be.sync):.2f} ms"
          f"  ({timed(lambda: be.stage_hash_rows(0, per, 0, local, 2 * (rows // G // 2) * 3, rows), be.sync):.2f} ms per group)")
# ---- the same rank with its rows dealt ROUND ROBIN in C ranges (LG_RELAY_ROUND_ROBIN(C)): per range, the evaluation and the hash of
# all 65 536 columns (one plane group: the wrap-around ring) -- the pieces of the projection printed at the end
t_int = timed(lambda: be.stage_interpolate(None, 0, local), be.sync)
rr = {}
for C in (2, 4, 8):
    span = 2 * (local // C // 2)
    t_ev = timed(lambda: be.stage_evaluate_rows(0, span), be.sync)
    t_h = timed(lambda: be.stage_hash_rows(0, be.nplanes, 0, span, 2 * (rows // G // 2) * 3, rows), be.sync)
    t_hg = timed(lambda: be.stage_hash_rows(0, be.nplanes // 4, 0, span, 2 * (rows // G // 2) * 3, rows), be.sync)
    rr[C] = (span, t_ev, t_h, t_hg)
    print(f"relay rank, round robin {C} ranges: one range of {span} rows: evaluate {t_ev:.2f} ms, hash of all columns (resume + park) {t_h:.2f} ms, "
          f"of one of 4 plane groups (16 384 columns, four lanes per column) {t_hg:.2f} ms")
per = be.nplanes // 4
t_group = timed(lambda: be.stage_hash_rows(0, per, 0, local, 2 * (rows // G // 2) * 3, rows), be.sync)
t_hash_all = timed(lambda: be.stage_hash_rows(0, be.nplanes, 0, local, 2 * (rows // G // 2) * 3, rows), be.sync)
print(f"relay rank: one group of 4 planes resumed at an ODD row (the hand-over inside a Blake2s block: one lane per column): "
      f"{timed(lambda: be.stage_hash_rows(0, per, 0, local, 5017, rows), be.sync):.2f} ms -- why lg_relay_row_ranges cuts on even rows")
be.close()

# ---- coset: 2 planes, all rows (needs the coefficient rows of all rows: interpolate them here)
sb = HipStageBackend(rows, k, device=0, world=G, rank=3)
full = bench.synthetic_preenc(bench.LARGE_SEED, rows * k).reshape(rows, k, 4)
sb.stage_interpolate(full, 0, rows)
del full
planes = list(range(6, 8))
sb.stage_evaluate_rows(planes, 0, rows)
sb.sync()
print(f"coset rank: evaluate 2 planes x {rows} rows: {timed(lambda: sb.stage_evaluate_rows(planes, 0, rows), sb.sync):.2f} ms")
print(f"coset rank: hash 8192 columns x {rows} rows (four lanes per column, state carried): {timed(lambda: sb.stage_hash(planes), sb.sync):.2f} ms")
print(f"coset rank: evaluate + hash, chunk-pipelined (lg_stage_evaluate_hash): {timed(lambda: sb.stage_evaluate_hash(planes), sb.sync):.2f} ms")
sb.close()

# ---- PROJECTION of an 8-GPU S22 commit from the pieces measured above (ONE GPU; the hops' wire time -- 5.2 MB over xGMI, tens of
# microseconds -- and the concurrency of eight real devices are NOT measured: this is arithmetic on measured pieces, not a measurement)
#   contiguous, P plane groups : every rank encodes its rows at once (t_enc), then G + P - 1 steps of one group's hash each
#   round robin, C ranges      : rank 0 evaluates its first range, the chain of C * G hops (one range's hash each) starts, each
#                                rank's later ranges are evaluated beside the hops: the chain is ready to take a range as soon as
#                                both its evaluation and the previous hop are done
print()
print(f"projection, 8 GPUs, contiguous with 4 plane groups : {t_enc:.2f} + {G + 4 - 1} x {t_group:.2f} = {t_enc + (G + 3) * t_group:.1f} ms")
print(f"projection, 8 GPUs, contiguous with 1 plane group  : {t_enc:.2f} + {G} x {t_hash_all:.2f} = {t_enc + G * t_hash_all:.1f} ms")
for C, (span, t_ev, t_h, t_hg) in rr.items():
    # every rank: interpolate (t_int), then evaluate range after range; hop j of the chain (range j // G of rank j % G), plane group p,
    # starts when that range is evaluated, hop j - 1 has handed group p over and this rank has finished group p - 1
    for P, th in ((1, t_h), (4, t_hg)):
        done = [[0.0] * P for _ in range(C * G)]
        for j in range(C * G):
            ready = t_int + (j // G + 1) * t_ev               # the owner has evaluated its (j // G)-th range
            for p in range(P):
                done[j][p] = max(ready, done[j - 1][p] if j else 0.0, done[j][p - 1] if p else 0.0) + th
        print(f"projection, 8 GPUs, round robin {C} ranges per rank, {P} plane group(s): {C * G} hops x {P} x {th:.2f} ms behind the evaluations "
              f"({t_int:.2f} + {C} x {t_ev:.2f} ms per rank) = {done[-1][-1]:.1f} ms")
