"""Multi-GPU host layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  Two modes:

* `ShardedBatchCommitter` -- independent proofs are dealt to ranks; no data-path collective, only
  the 32-byte roots are gathered.  This is what `bench.py --gpus N` measures (BASELINE configs[4]).
* `CosetShardedCommitter` -- ONE large proof on G GPUs (BASELINE configs[3]).  A column's Blake2s
  chain spans all rows, so rows cannot be sharded end to end; instead
    1. rank g interpolates its row shard                                   (mod.rs:521-526)
    2. ONE in-place all-gather of the coefficient rows (bulk: 4m*k*32 B)   RCCL
       (row shards are ceil(rows/G) rows each, the coefficient buffer is padded to G such
       shards, so the shapes BASELINE names -- 20 068 and 10 036 rows on 8 GPUs -- need no
       ragged fallback; the last rank's shard is short and its padding rows are never read)
    3. rank g evaluates + hashes the coset planes it owns, for ALL rows    (mod.rs:528-542)
    4. all-gather of the n 32-byte leaf digests (tiny)                     RCCL
    5. every rank builds the (replicated) tree                             (mod.rs:544-551)
  Column j = np*q + s is opened by the owner of plane s.

The device work goes through a *backend* object (`HipStageBackend` wraps the C ABI's staged
calls and exposes the resident buffers as torch tensors without copies); the CPU tests inject
an oracle-backed stand-in with the same methods, so the orchestration below is exercised with
gloo at world_size 2 without a GPU.
"""
from __future__ import annotations

import ctypes
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi

_vp = ctypes.c_void_p


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous, balanced [begin, end) of `total` items for `rank` (proofs in throughput mode)"""
    return (total * rank) // world, (total * (rank + 1)) // world


def padded_shard_rows(total: int, world: int) -> int:
    """rows per rank when `total` rows are dealt in `world` EQUAL shards (the last one padded)"""
    return -(-total // world)


def padded_shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """real rows [begin, end) of `rank`'s shard of ceil(total / world) rows (empty for trailing ranks of tiny inputs)"""
    per = padded_shard_rows(total, world)
    return min(total, rank * per), min(total, (rank + 1) * per)


def owned_planes(nplanes: int, world: int, rank: int) -> List[int]:
    """planes are dealt in contiguous equal runs; nplanes must be divisible by world"""
    if nplanes % world != 0:
        raise ValueError(f"{nplanes} coset planes cannot be split evenly over {world} ranks")
    per = nplanes // world
    return list(range(rank * per, (rank + 1) * per))


class _CudaArray:
    """minimal __cuda_array_interface__ holder so torch can alias a raw device pointer"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class HipStageBackend:
    """Staged single-proof commit on this rank's GPU through the C ABI (include/ligero_hip.h:
    lg_stage_interpolate / lg_stage_evaluate_hash / lg_stage_merkle / lg_device_buffer)."""

    def __init__(self, rows: int, k: int, device: int = 0, world: int = 1, rank: int = 0):
        """One rank of `world`: the context allocates only this rank's coset planes of U and a coefficient buffer
        padded to `world` equal row shards (lg_ctx_create_sharded)."""
        from .ligero import LigeroCommitter
        self.rows, self.k, self.n, self.device = rows, k, 8 * k, device
        self.nplanes = 8 if k <= 4096 else 8 * (k // 4096)
        planes = owned_planes(self.nplanes, world, rank)
        self.coeff_rows = world * padded_shard_rows(rows, world)
        self.c = LigeroCommitter(rows=rows, k=k, batch=1, device=device, shard=(planes[0], len(planes), self.coeff_rows))
        assert self.c.planes() == (self.nplanes, planes[0], len(planes))
        self._L = _ffi.lib()

    def _buffer(self, which: int):
        import torch
        ptr, size = _vp(), ctypes.c_size_t()
        _ffi.check(self._L.lg_device_buffer(self.c._ctx, which, ctypes.cast(ctypes.byref(ptr), _vp), ctypes.cast(ctypes.byref(size), _vp)),
                   "lg_device_buffer", self.c._ctx)
        return torch.as_tensor(_CudaArray(ptr.value, size.value), device=f"cuda:{self.device}")

    def stage_interpolate(self, preenc_rows: Optional[np.ndarray], row0: int, nrows: int):
        p = None
        if preenc_rows is not None:
            preenc_rows = np.ascontiguousarray(preenc_rows, dtype=np.uint64)
            assert preenc_rows.size == nrows * self.k * 4
            p = preenc_rows.ctypes.data_as(_vp)
        _ffi.check(self._L.lg_stage_interpolate(self.c._ctx, p, row0, nrows), "lg_stage_interpolate", self.c._ctx)

    def stage_evaluate_hash(self, planes: Sequence[int]):
        mask = 0
        for s in planes:
            mask |= 1 << s
        _ffi.check(self._L.lg_stage_evaluate_hash(self.c._ctx, mask), "lg_stage_evaluate_hash", self.c._ctx)

    @staticmethod
    def _mask(planes: Sequence[int]) -> int:
        mask = 0
        for s in planes:
            mask |= 1 << s
        return mask

    def stage_evaluate_rows(self, planes: Sequence[int], row0: int, nrows: int):
        """the planes' evaluation of rows [row0, row0 + nrows) only (any rows, any order; lg_stage_evaluate_rows)"""
        _ffi.check(self._L.lg_stage_evaluate_rows(self.c._ctx, self._mask(planes), row0, nrows), "lg_stage_evaluate_rows", self.c._ctx)

    def stage_hash(self, planes: Sequence[int]):
        """the planes' column hashes over all rows, once every row is evaluated (lg_stage_hash)"""
        _ffi.check(self._L.lg_stage_hash(self.c._ctx, self._mask(planes)), "lg_stage_hash", self.c._ctx)

    def stage_merkle(self):
        _ffi.check(self._L.lg_stage_merkle(self.c._ctx), "lg_stage_merkle", self.c._ctx)

    def sync(self):
        self.c.sync()

    def coeffs_bytes(self):
        """[coeff_rows, k*32] uint8 view of the resident coefficient rows (coeff_rows = world * ceil(rows / world):
        rows past `rows` are all-gather padding)"""
        return self._buffer(_ffi.LG_BUF_COEFFS).view(self.coeff_rows, self.k * 32)

    def leaves_bytes(self):
        """[n, 32] uint8 view of the resident leaf digests"""
        return self._buffer(_ffi.LG_BUF_LEAVES).view(self.n, 32)

    def digests_pack(self, world: int, rank: int):
        """lg_stage_digests_pack: this rank's leaf digests into block `rank` of the staging buffer -> [world, block] uint8 view of
        it (all-gather it in place after sync())"""
        import torch
        ptr, size = _vp(), ctypes.c_size_t()
        _ffi.check(self._L.lg_stage_digests_pack(self.c._ctx, world, rank, ctypes.cast(ctypes.byref(ptr), _vp), ctypes.cast(ctypes.byref(size), _vp)),
                   "lg_stage_digests_pack", self.c._ctx)
        return torch.as_tensor(_CudaArray(ptr.value, world * size.value), device=f"cuda:{self.device}").view(world, size.value)

    def digests_unpack(self, world: int):
        _ffi.check(self._L.lg_stage_digests_unpack(self.c._ctx, world), "lg_stage_digests_unpack", self.c._ctx)

    def root(self) -> bytes:
        return self.c.root()

    def open_columns(self, indices):
        return self.c.open_columns(indices)

    def close(self):
        self.c.close()


class CosetShardedCommitter:
    """One proof over `world` ranks.  `backend` does the device work; `dist` is torch.distributed
    (already initialised) or None for a single process."""

    def __init__(self, backend, dist=None, group=None, collectives_at_world_1: bool = False, exchange_pieces: int = 1):
        """collectives_at_world_1: issue the two all-gathers even in a one-rank group (they are identities then) -- lets a
        one-GPU box run the exact RCCL calls of the multi-GPU path (bench.py LIGERO_BENCH_FORCE_DIST, tests).
        exchange_pieces > 1: the coefficient all-gather is cut into that many pieces (piece c = the c-th slice of EVERY rank's
        shard, so every piece uses all links), issued asynchronously, and the evaluation of piece c runs while piece c + 1
        is on the wire; the column hash, which needs the rows in order, follows the last piece (commit_pipelined)."""
        self.be = backend
        self.dist = dist
        self.group = group
        self.pieces = max(1, int(exchange_pieces))
        self._piece_buf = [None, None]
        self.force = bool(collectives_at_world_1) and dist is not None
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.planes = owned_planes(backend.nplanes, self.world, self.rank)
        self.shard_rows = padded_shard_rows(backend.rows, self.world)
        self.stage_ms: Dict[str, float] = {}      # wall time of each stage of the last commit() (each ends in a device sync)
        self._digest_buf = None

    def row_range(self, rank: Optional[int] = None) -> Tuple[int, int]:
        return padded_shard_range(self.be.rows, self.world, self.rank if rank is None else rank)

    def commit(self, preenc_rows_local: Optional[np.ndarray]) -> bytes:
        """preenc_rows_local: this rank's rows [row_range()) of preenc_u (None: they are resident from an earlier
        commit).  Returns u_root."""
        if self.pieces > 1 and (self.world > 1 or self.force):
            return self.commit_pipelined(preenc_rows_local)
        be, dist = self.be, self.dist
        ms = self.stage_ms = {}
        r0, r1 = self.row_range()
        t = time.perf_counter()

        def lap(name):
            nonlocal t
            now = time.perf_counter()
            ms[name] = (now - t) * 1e3
            t = now

        be.stage_interpolate(preenc_rows_local, r0, r1 - r0)
        be.sync()
        lap("interpolate")
        if self.world > 1 or self.force:
            # equal (padded) shards: ONE in-place all-gather whatever rows % world is
            coeffs = be.coeffs_bytes()
            p0 = self.rank * self.shard_rows
            dist.all_gather_into_tensor(coeffs.view(-1), coeffs[p0:p0 + self.shard_rows].view(-1), group=self.group)
            self._device_sync(coeffs)
        lap("allgather_coeffs")
        be.stage_evaluate_hash(self.planes)
        be.sync()
        lap("evaluate_hash")
        return self._digests_and_tree(ms)

    def _digests_and_tree(self, ms: Dict[str, float]) -> bytes:
        """steps 4 and 5: all-gather of the leaf digests, replicated tree"""
        be = self.be
        t = time.perf_counter()
        if self.world > 1 or self.force:
            import torch
            np_, per = be.nplanes, len(self.planes)
            leaves = be.leaves_bytes().view(be.n // np_, self.world, per, 32)   # [q][owner][plane of owner][32]
            mine = leaves[:, self.rank].contiguous()
            if self._digest_buf is None or self._digest_buf.device != mine.device:
                self._digest_buf = torch.empty((self.world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            self.dist.all_gather_into_tensor(self._digest_buf.view(-1), mine.view(-1), group=self.group)   # flat: gloo insists on 1-D shapes
            leaves.copy_(self._digest_buf.permute(1, 0, 2, 3))                  # one strided copy back into leaf order
            self._device_sync(leaves)
        ms["allgather_digests"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        be.stage_merkle()
        be.sync()
        ms["merkle"] = (time.perf_counter() - t) * 1e3
        return be.root()

    def piece_plan(self) -> List[Tuple[int, int]]:
        """[(first row inside a shard, rows)] of the exchange pieces: equal slices of the padded shard, the last one short"""
        per = -(-self.shard_rows // min(self.pieces, max(1, self.shard_rows)))
        return [(o, min(per, self.shard_rows - o)) for o in range(0, self.shard_rows, per)]

    def commit_pipelined(self, preenc_rows_local: Optional[np.ndarray]) -> bytes:
        """commit() with the coefficient all-gather hidden behind the evaluation.  Piece c of the exchange is rows
        [o_c, o_c + n_c) of every rank's shard: one all_gather_into_tensor into a staging buffer (two, alternating) and one
        strided copy into LG_BUF_COEFFS; its rows are then evaluated (lg_stage_evaluate_rows, one call per source shard, padding
        rows of the last shard skipped) while the next piece is in flight.  stage_ms: `allgather_coeffs` is the time this
        rank spent WAITING for pieces, `evaluate_hash` the rest of the loop plus the hash."""
        import torch
        be, dist = self.be, self.dist
        ms = self.stage_ms = {}
        r0, r1 = self.row_range()
        t = time.perf_counter()
        be.stage_interpolate(preenc_rows_local, r0, r1 - r0)
        be.sync()
        ms["interpolate"] = (time.perf_counter() - t) * 1e3
        coeffs = be.coeffs_bytes()                                   # [world * shard_rows, k * 32]
        width = coeffs.shape[1]
        shards = coeffs.view(self.world, self.shard_rows, width)
        plan = self.piece_plan()
        cap = plan[0][1]
        for i in (0, 1):
            if self._piece_buf[i] is None or self._piece_buf[i].device != coeffs.device or self._piece_buf[i].shape != (self.world, cap, width):
                self._piece_buf[i] = torch.empty((self.world, cap, width), dtype=coeffs.dtype, device=coeffs.device)

        def start(c):
            o, n = plan[c]
            out = self._piece_buf[c & 1][:, :n, :] if n == cap else self._piece_buf[c & 1].view(-1)[:self.world * n * width].view(self.world, n, width)
            mine = shards[self.rank, o:o + n, :]
            return out, dist.all_gather_into_tensor(out.reshape(-1), mine.reshape(-1), group=self.group, async_op=True)

        wait_s = 0.0
        t_loop = time.perf_counter()
        pending = start(0)
        for c, (o, n) in enumerate(plan):
            out, work = pending
            tw = time.perf_counter()
            work.wait()                                              # torch's current stream now follows the collective ...
            self._stream_sync(coeffs)                                # ... and the host follows that stream (NOT the library's streams,
            wait_s += time.perf_counter() - tw                       #     where the previous piece may still be being evaluated)
            # every other rank's slice goes to its place (this rank's own rows are already there)
            for g in range(self.world):
                if g != self.rank:
                    shards[g, o:o + n, :].copy_(out[g])
            self._stream_sync(coeffs)                                # the library's stream may read them from here on
            if c + 1 < len(plan):
                pending = start(c + 1)                               # on the wire while the rows below are evaluated
            for g in range(self.world):
                a = g * self.shard_rows + o
                b = min(a + n, be.rows)                              # the last shard is short: its padding rows are never evaluated
                if b > a:
                    be.stage_evaluate_rows(self.planes, a, b - a)
        be.stage_hash(self.planes)
        be.sync()
        ms["allgather_coeffs"] = wait_s * 1e3
        ms["evaluate_hash"] = (time.perf_counter() - t_loop - wait_s) * 1e3
        return self._digests_and_tree(ms)

    @staticmethod
    def _stream_sync(t):
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    @staticmethod
    def _device_sync(t):
        if t.is_cuda:
            import torch
            torch.cuda.synchronize(t.device)

    def open_columns(self, indices: Sequence[int]):
        """Each rank opens the columns whose plane it owns; returns {index: (column, sibling, path)}
        for those (the caller merges ranks with all_gather_object if it needs them in one place)."""
        np_ = self.be.nplanes
        mine = [int(j) for j in indices if (int(j) % np_) in self.planes]
        if not mine:
            return {}
        cols, sib, paths = self.be.open_columns(mine)
        return {j: (cols[i], sib[i], paths[i]) for i, j in enumerate(mine)}


# ---------------------------------------------------------------------------------------------- row-relay mode
def relay_row_ranges(rows: int, world: int, rank: int, layout: str = "contiguous") -> List[Tuple[int, int]]:
    """[(first row in the column, rows)] this rank keeps in row-relay mode, in column order (empty ranges dropped).
    "contiguous": one balanced range per rank.  "blocks": the rank's share of each of the four row blocks X, Y, Z, W of
    preenc_u (mod.rs:516; rows = 4m) -- its rows then form a small [X; Y; Z; W] matrix of their own, which is what the
    quadratic test's row triples (x_i, y_i, z_i) need to stay on one rank."""
    if layout == "contiguous":
        a, b = shard_range(rows, world, rank)
        return [(a, b - a)] if b > a else []
    if layout != "blocks":
        raise ValueError(f"unknown relay layout {layout!r}")
    if rows % 4:
        raise ValueError("the block layout needs rows = 4 m")
    m = rows // 4
    a, b = shard_range(m, world, rank)
    return [(blk * m + a, b - a) for blk in range(4)] if b > a else []


def relay_chain(rows: int, world: int, layout: str = "contiguous") -> List[Tuple[int, int, int, int]]:
    """every range of every rank in column order: [(first row, rows, owner rank, first row inside the owner's matrix)]"""
    chain = []
    for r in range(world):
        local = 0
        for pos, n in relay_row_ranges(rows, world, r, layout):
            chain.append((pos, n, r, local))
            local += n
    chain.sort()
    assert sum(n for _, n, _, _ in chain) == rows and all(chain[i][0] + chain[i][1] == chain[i + 1][0] for i in range(len(chain) - 1))
    return chain


class HipRelayBackend:
    """One rank of a row-relay commit on this rank's GPU: an ordinary batch-1 context of the rank's OWN row count
    (include/ligero_hip.h: lg_stage_interpolate / lg_stage_evaluate_rows / lg_stage_hash_rows / lg_stage_merkle)."""

    def __init__(self, local_rows: int, k: int, device: int = 0):
        from .ligero import LigeroCommitter
        self.local_rows, self.k, self.n, self.device = local_rows, k, 8 * k, device
        self.nplanes = 8 if k <= 4096 else 8 * (k // 4096)
        self.ki = self.n // self.nplanes
        # a rank without rows still takes part in the broadcast of the digests and builds the tree
        self.c = LigeroCommitter(rows=max(1, local_rows), k=k, batch=1, device=device)
        self._L = _ffi.lib()
        self._all = (1 << self.nplanes) - 1

    def _buffer(self, which: int):
        import torch
        ptr, size = _vp(), ctypes.c_size_t()
        _ffi.check(self._L.lg_device_buffer(self.c._ctx, which, ctypes.cast(ctypes.byref(ptr), _vp), ctypes.cast(ctypes.byref(size), _vp)),
                   "lg_device_buffer", self.c._ctx)
        return torch.as_tensor(_CudaArray(ptr.value, size.value), device=f"cuda:{self.device}")

    def stream(self):
        """the library's stream as a torch stream: collectives issued under it are ordered with the library's calls"""
        import torch
        ptr = _vp()
        _ffi.check(self._L.lg_ctx_stream(self.c._ctx, ctypes.cast(ctypes.byref(ptr), _vp)), "lg_ctx_stream", self.c._ctx)
        return torch.cuda.ExternalStream(ptr.value, device=f"cuda:{self.device}")

    def pipeline_chunks(self) -> int:
        return self.c.pipeline_chunks()

    def stage_interpolate(self, preenc_rows: Optional[np.ndarray], row0: int, nrows: int):
        p = None
        if preenc_rows is not None:
            preenc_rows = np.ascontiguousarray(preenc_rows, dtype=np.uint64)
            assert preenc_rows.size == nrows * self.k * 4
            p = preenc_rows.ctypes.data_as(_vp)
        _ffi.check(self._L.lg_stage_interpolate(self.c._ctx, p, row0, nrows), "lg_stage_interpolate", self.c._ctx)

    def stage_evaluate_rows(self, row0: int, nrows: int):
        _ffi.check(self._L.lg_stage_evaluate_rows(self.c._ctx, self._all, row0, nrows), "lg_stage_evaluate_rows", self.c._ctx)

    def stage_hash_rows(self, plane0: int, nplanes: int, row0: int, nrows: int, col_pos: int, col_rows: int):
        mask = ((1 << nplanes) - 1) << plane0
        _ffi.check(self._L.lg_stage_hash_rows(self.c._ctx, mask, row0, nrows, col_pos, col_rows), "lg_stage_hash_rows", self.c._ctx)

    def hstate_bytes(self):
        """[nplanes, ki * LG_HSTATE_BYTES] uint8 view of the parked Blake2s states (settles the hash stream)"""
        return self._buffer(_ffi.LG_BUF_HSTATE).view(self.nplanes, self.ki * _ffi.LG_HSTATE_BYTES)

    def leaves_bytes(self):
        return self._buffer(_ffi.LG_BUF_LEAVES).view(self.n, 32)

    def stage_merkle(self):
        _ffi.check(self._L.lg_stage_merkle(self.c._ctx), "lg_stage_merkle", self.c._ctx)

    def sync(self):
        self.c.sync()

    def root(self) -> bytes:
        return self.c.root()

    def open_columns(self, indices):
        """this rank's ROWS of the opened columns, the siblings and the (complete) paths"""
        if self.local_rows == 0:
            # a rank without rows holds the (replicated) tree only: Path::leaf_sibling_hash and auth_path (root side first) read
            # off the heap-ordered nodes, as lg_open_columns does on the device
            leaves, nodes = self.c.leaves()[0], self.c.nodes()[0]
            logn = self.n.bit_length() - 1
            idx = [int(j) for j in indices]
            sib = np.stack([leaves[j ^ 1] for j in idx]) if idx else np.empty((0, 32), dtype=np.uint8)
            paths = np.empty((len(idx), logn - 1, 32), dtype=np.uint8)
            for c, j in enumerate(idx):
                for depth in range(1, logn):
                    paths[c, depth - 1] = nodes[((1 << depth) - 1) + ((j >> (logn - depth)) ^ 1)]
            return np.empty((len(idx), 0, 4), dtype=np.uint64), sib, paths
        cols, sib, paths = self.c.open_columns(indices)
        return cols[:, :self.local_rows], sib, paths

    def close(self):
        self.c.close()


class RowRelayCommitter:
    """ONE proof over `world` ranks, rows sharded END TO END (BASELINE.json north_star: "rows shard naturally ... all-gather
    of column digests"): rank g interpolates and evaluates every coset plane of its own rows and keeps them; a column's
    Blake2s (mod.rs:536-542) absorbs the rows in order, so the ranks take turns and the 72-byte state of every column is
    handed from rank to rank -- n * 80 bytes per hop instead of the 4m * k * 32-byte coefficient all-gather of the
    coset-sharded mode.  The rank with the last rows broadcasts the n digests and every rank builds the tree.

    `make_backend(local_rows)` builds this rank's backend (HipRelayBackend; the CPU tests inject an oracle-backed double).
    plane_groups P > 1 cuts every hop into P runs of planes: rank g works on group c while rank g + 1 works on group c - 1.
    That pays only if a hash launch over fewer columns is faster -- on MI355X it is not (one lane per column, one wave per SIMD
    at most: a latency chain whose length does not depend on the number of columns) -- so the default is 1.
    layout: relay_row_ranges()."""

    def __init__(self, make_backend, rows: int, dist=None, group=None, plane_groups: int = 1, layout: str = "contiguous",
                 collectives_at_world_1: bool = False):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.rows, self.layout = rows, layout
        self.chain = relay_chain(rows, self.world, layout)
        self.mine = [(pos, n, local) for pos, n, r, local in self.chain if r == self.rank]
        self.local_rows = sum(n for _, n, _ in self.mine)
        self.be = make_backend(self.local_rows)
        self.groups = max(1, min(int(plane_groups), self.be.nplanes))
        while self.be.nplanes % self.groups:
            self.groups -= 1
        if layout != "contiguous":
            # the block layout's chain wraps around (rank G - 1 hands back to rank 0): with several groups in flight a rank
            # would have to send and receive in the same step, which plain blocking sends cannot express; one group = a strictly
            # sequential chain, which is also the faster choice (see the class comment)
            self.groups = 1
        self.force = bool(collectives_at_world_1) and dist is not None
        self.stage_ms: Dict[str, float] = {}
        self._stream = self.be.stream() if hasattr(self.be, "stream") else None
        self._nccl = dist is not None and dist.get_backend(group) == "nccl"

    def row_ranges(self, rank: Optional[int] = None) -> List[Tuple[int, int]]:
        return relay_row_ranges(self.rows, self.world, self.rank if rank is None else rank, self.layout)

    # -- the three transfers; device tensors go through RCCL as they are, any other backend gets host copies
    def _send(self, t, dst):
        self.dist.send(t if self._nccl or not t.is_cuda else t.cpu(), dst, group=self.group)

    def _recv(self, t, src):
        if self._nccl or not t.is_cuda:
            self.dist.recv(t, src, group=self.group)
        else:
            h = t.cpu()
            self.dist.recv(h, src, group=self.group)
            t.copy_(h)

    def _broadcast(self, t, src):
        if self._nccl or not t.is_cuda:
            self.dist.broadcast(t, src, group=self.group)
        else:
            h = t.cpu()
            self.dist.broadcast(h, src, group=self.group)
            if self.rank != src:
                t.copy_(h)

    def commit(self, preenc_rows_local: Optional[np.ndarray]) -> bytes:
        """preenc_rows_local: this rank's rows (its ranges, concatenated in column order), or None when they are resident from
        an earlier commit.  Returns u_root."""
        import contextlib
        import torch
        be = self.be
        ctx = torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()
        marks = []

        def mark(name):
            if self._stream is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(self._stream)
                marks.append((name, e))
            else:
                marks.append((name, time.perf_counter()))

        with ctx:
            mark("start")
            per, gp = be.nplanes // self.groups, self.groups
            head_done = False
            if self.local_rows:
                be.stage_interpolate(preenc_rows_local, 0, self.local_rows)
                # evaluate in row chunks; the rank that holds the first rows of the columns hashes each chunk as soon as it is
                # evaluated (the library queues the hash on its second stream, beside the evaluation of the next chunk)
                pos0, n0, local0 = self.mine[0]
                nch = max(1, min(be.pipeline_chunks() if hasattr(be, "pipeline_chunks") else 1, n0))
                head = pos0 == 0
                for c in range(nch):
                    a, b = (n0 * c) // nch, (n0 * (c + 1)) // nch
                    be.stage_evaluate_rows(local0 + a, b - a)
                    if head:
                        be.stage_hash_rows(0, be.nplanes, local0 + a, b - a, pos0 + a, self.rows)
                head_done = head
                rest0 = local0 + n0
                if self.local_rows > rest0:
                    be.stage_evaluate_rows(rest0, self.local_rows - rest0)
            mark("encode")
            exchange = self.world > 1
            # the relay: every range of the chain in column order, every plane group of it in turn
            for i, (pos, n, owner, local) in enumerate(self.chain):
                prev_owner = self.chain[i - 1][2] if i > 0 else None
                next_owner = self.chain[i + 1][2] if i + 1 < len(self.chain) else None
                if owner != self.rank:
                    continue
                for g in range(gp):
                    if exchange and prev_owner is not None and prev_owner != self.rank:
                        self._recv(be.hstate_bytes()[g * per:(g + 1) * per], prev_owner)
                    if not (head_done and i == 0):
                        be.stage_hash_rows(g * per, per, local, n, pos, self.rows)
                    if exchange and next_owner is not None and next_owner != self.rank:
                        self._send(be.hstate_bytes()[g * per:(g + 1) * per], next_owner)
            mark("relay")
            if exchange or self.force:
                self._broadcast(be.leaves_bytes(), self.chain[-1][2] if exchange else 0)
            mark("digests")
            be.stage_merkle()
            mark("merkle")
        be.sync()
        ms = self.stage_ms = {}
        for (_, a), (name, b) in zip(marks, marks[1:]):
            ms[name] = a.elapsed_time(b) if self._stream is not None else (b - a) * 1e3
        return be.root()

    def open_columns(self, indices: Sequence[int]):
        """every rank holds ITS ROWS of every column: returns (rows of the columns [t, local_rows, 4] in this rank's own row
        order, leaf siblings, authentication paths); assemble_columns() merges the ranks' pieces"""
        return self.be.open_columns([int(j) for j in indices])

    def assemble_columns(self, pieces: Sequence[np.ndarray]) -> np.ndarray:
        """pieces[r] = rank r's rows of the opened columns ([t, local_rows_r, 4]) -> the columns [t, rows, 4] (u.column(j),
        src/matrices/mod.rs:169-171)"""
        t = pieces[0].shape[0]
        out = np.empty((t, self.rows, 4), dtype=np.uint64)
        for pos, n, owner, local in self.chain:
            out[:, pos:pos + n] = pieces[owner][:, local:local + n]
        return out


class ShardedBatchCommitter:
    """Independent proofs dealt to ranks (weak scaling, no data-path collective).
    `make_committer(batch_local)` builds this rank's committer (a `LigeroCommitter` on the GPU)."""

    def __init__(self, make_committer, batch: int, dist=None, group=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.batch = batch
        self.b0, self.b1 = shard_range(batch, self.world, self.rank)
        self.c = make_committer(self.b1 - self.b0) if self.b1 > self.b0 else None

    def commit(self, preenc_local) -> bytes:
        """preenc_local: the rows of proofs [b0, b1).  Returns all `batch` roots, concatenated in
        proof order, on every rank."""
        local = b""
        if self.c is not None:
            _, local = self.c.encode_commit(preenc_local, want_coeffs=False)
        if self.world == 1:
            return local
        gathered = [None] * self.world
        self.dist.all_gather_object(gathered, local, group=self.group)   # 32 bytes per proof: control plane only
        return b"".join(gathered)
