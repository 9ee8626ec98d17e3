"""Several sharded contexts in ONE process on one GPU, the two all-gathers of the coset-sharded commit done as device copies
between them (tests/test_gpu_sharded_subproofs.py, tests/parity_sweep.py): every C-ABI call of the multi-GPU path runs without a
process group."""
import numpy as np


def sharded_commit(backends, pre):
    """the five stages of CosetShardedCommitter.commit with the exchanges done as device copies between the contexts"""
    import torch
    world = len(backends)
    rows = backends[0].rows
    per_rows = -(-rows // world)
    for r, be in enumerate(backends):
        r0, r1 = min(rows, r * per_rows), min(rows, (r + 1) * per_rows)
        be.stage_interpolate(pre[r0:r1] if r1 > r0 else None, r0, r1 - r0)
        be.sync()
    for dst in backends:                                                   # all-gather of the coefficient rows
        for r, src in enumerate(backends):
            if src is not dst:
                dst.coeffs_bytes()[r * per_rows:(r + 1) * per_rows].copy_(src.coeffs_bytes()[r * per_rows:(r + 1) * per_rows])
    torch.cuda.synchronize()
    per_planes = backends[0].nplanes // world
    for r, be in enumerate(backends):
        be.stage_evaluate_hash(range(r * per_planes, (r + 1) * per_planes))
        be.sync()
    bufs = []
    for r, be in enumerate(backends):                                      # pack, "all-gather in place", unpack
        bufs.append(be.digests_pack(world, r))
        be.sync()
    for dst_r, dst in enumerate(bufs):
        for r, src in enumerate(bufs):
            if r != dst_r:
                dst[r].copy_(src[r])
    torch.cuda.synchronize()
    roots = []
    for be in backends:
        be.digests_unpack(world)
        be.stage_merkle()
        be.sync()
        roots.append(be.root())
    return roots


def merge_points(points_and_masks, nplanes):
    """slot j of the 2k-slot array belongs to plane 4 (j mod np/4): take it from the context that served that plane"""
    out = np.zeros_like(points_and_masks[0][0])
    owner = {}
    for i, (_, mask) in enumerate(points_and_masks):
        for s in range(nplanes):
            if mask & (1 << s):
                assert s not in owner, "two contexts served the same plane"
                owner[s] = i
    for j in range(out.shape[0]):
        s = 4 * (j % (nplanes // 4))
        if s in owner:
            out[j] = points_and_masks[owner[s]][0][j]
    return out, owner


def relay_commit(backends, chain, pre, rows):
    """the row-relay commit (ligero_amd.sharded.RowRelayCommitter) over several HipRelayBackend contexts of ONE process: every
    hop of the column states and the broadcast of the digests as a device copy.  chain: relay_chain(rows, world, layout)."""
    import torch
    for r, be in enumerate(backends):
        mine = [(pos, n) for pos, n, owner, _ in chain if owner == r]
        if be.local_rows:
            be.stage_interpolate(np.concatenate([pre[a:a + n] for a, n in mine]), 0, be.local_rows)
            be.stage_evaluate_rows(0, be.local_rows)
            be.sync()
    prev = None
    for pos, n, owner, local in chain:
        be = backends[owner]
        if prev is not None and prev is not be:
            prev.sync()
            be.hstate_bytes().copy_(prev.hstate_bytes())
            torch.cuda.synchronize()
        be.stage_hash_rows(0, be.nplanes, local, n, pos, rows)
        prev = be
    prev.sync()
    roots = []
    for be in backends:
        if be is not prev:
            be.leaves_bytes().copy_(prev.leaves_bytes())
            torch.cuda.synchronize()
        be.stage_merkle()
        be.sync()
        roots.append(be.root())
    return roots
