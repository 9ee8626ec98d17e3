"""Resumable column hash for the row-relay commit -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).

The reference hashes one column with one call: Blake2s-256 over serialize_compressed(column) =
LE64(rows) || 32-byte canonical elements (/root/reference/src/ligero/mod.rs:536-542 with
H = FieldToBytesColHasher<F, Blake2s256>, src/ligero/types.rs:18).  When the rows of a proof are
sharded over several GPUs every rank absorbs its own rows of every column in turn and hands the
state on.  This module restates that in numpy, vectorised over the columns: `ColumnRelayHasher`
absorbs row ranges at any row position, exports / imports the parked state in the layout of the
device library's LG_BUF_HSTATE record (include/ligero_hip.h: 32 bytes of chaining value, then
the bytes of the 64-byte block in progress -- 8 after an even number of rows, 40 after an odd
one -- padded to 80), and finalises to the digests `hashlib.blake2s` gives for the whole column
(tests/test_oracle.py pins that equality; RFC 7693 is the specification of the compression).
"""
from __future__ import annotations

import numpy as np

HSTATE_BYTES = 80
_IV = np.array([0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19], dtype=np.uint32)
_SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]


def _rotr(x, n):
    return (x >> np.uint32(n)) | (x << np.uint32(32 - n))


def compress(h: np.ndarray, m: np.ndarray, t: int, last: bool) -> np.ndarray:
    """RFC 7693 section 3.2 for many independent states at once: h (C, 8) uint32, m (C, 16) uint32 -> new h"""
    v = [h[:, i].copy() for i in range(8)] + [np.full(h.shape[0], _IV[i], dtype=np.uint32) for i in range(8)]
    v[12] = v[12] ^ np.uint32(t & 0xFFFFFFFF)
    v[13] = v[13] ^ np.uint32((t >> 32) & 0xFFFFFFFF)
    if last:
        v[14] = ~v[14]

    def g(a, b, c, d, x, y):
        v[a] = v[a] + v[b] + x
        v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = v[c] + v[d]
        v[b] = _rotr(v[b] ^ v[c], 12)
        v[a] = v[a] + v[b] + y
        v[d] = _rotr(v[d] ^ v[a], 8)
        v[c] = v[c] + v[d]
        v[b] = _rotr(v[b] ^ v[c], 7)

    with np.errstate(over="ignore"):
        for r in range(10):
            s = _SIGMA[r]
            g(0, 4, 8, 12, m[:, s[0]], m[:, s[1]])
            g(1, 5, 9, 13, m[:, s[2]], m[:, s[3]])
            g(2, 6, 10, 14, m[:, s[4]], m[:, s[5]])
            g(3, 7, 11, 15, m[:, s[6]], m[:, s[7]])
            g(0, 5, 10, 15, m[:, s[8]], m[:, s[9]])
            g(1, 6, 11, 12, m[:, s[10]], m[:, s[11]])
            g(2, 7, 8, 13, m[:, s[12]], m[:, s[13]])
            g(3, 4, 9, 14, m[:, s[14]], m[:, s[15]])
    out = h.copy()
    for i in range(8):
        out[:, i] ^= v[i] ^ v[8 + i]
    return out


class ColumnRelayHasher:
    """Blake2s-256 of `columns` columns of `col_rows` 32-byte elements each, absorbed in row ranges."""

    def __init__(self, columns: int, col_rows: int):
        self.c, self.col_rows = columns, col_rows
        self.pos = 0                                    # rows absorbed so far
        self.h = np.tile(_IV, (columns, 1))
        self.h[:, 0] ^= np.uint32(0x01010020)           # digest length 32, no key, fanout = depth = 1
        self.t = 0                                      # bytes compressed so far
        self.buf = np.tile(np.frombuffer(int(col_rows).to_bytes(8, "little"), dtype=np.uint8), (columns, 1))   # LE64(rows)

    def absorb(self, rows_canonical_bytes: np.ndarray):
        """rows_canonical_bytes: (nrows, columns, 32) uint8 -- the canonical little-endian bytes of rows pos .. pos + nrows"""
        rows = np.ascontiguousarray(rows_canonical_bytes, dtype=np.uint8)
        assert rows.ndim == 3 and rows.shape[1] == self.c and rows.shape[2] == 32
        assert self.pos + rows.shape[0] <= self.col_rows
        for r in rows:
            self.buf = np.concatenate([self.buf, r], axis=1)
            if self.buf.shape[1] > 64:                  # a block is compressed only once a byte beyond it exists (the last one is special)
                self.t += 64
                self.h = compress(self.h, np.ascontiguousarray(self.buf[:, :64]).view("<u4"), self.t, False)
                self.buf = np.ascontiguousarray(self.buf[:, 64:])
        self.pos += rows.shape[0]

    def export_state(self) -> np.ndarray:
        """(columns, 80) uint8: the device's LG_BUF_HSTATE record of every column"""
        carry = 40 if self.pos & 1 else 8
        assert self.buf.shape[1] == carry and self.t == 64 * (self.pos // 2)
        out = np.zeros((self.c, HSTATE_BYTES), dtype=np.uint8)
        out[:, :32] = self.h.astype("<u4").view(np.uint8).reshape(self.c, 32)
        out[:, 32:32 + carry] = self.buf
        return out

    @classmethod
    def import_state(cls, state: np.ndarray, pos: int, col_rows: int) -> "ColumnRelayHasher":
        state = np.ascontiguousarray(state, dtype=np.uint8).reshape(-1, HSTATE_BYTES)
        x = cls(state.shape[0], col_rows)
        x.pos, x.t = pos, 64 * (pos // 2)
        x.h = np.ascontiguousarray(state[:, :32]).view("<u4").astype(np.uint32)
        x.buf = np.ascontiguousarray(state[:, 32:32 + (40 if pos & 1 else 8)])
        return x

    def digests(self) -> np.ndarray:
        """(columns, 32) uint8 once every row has been absorbed"""
        assert self.pos == self.col_rows
        last = np.zeros((self.c, 64), dtype=np.uint8)
        last[:, :self.buf.shape[1]] = self.buf
        h = compress(self.h, last.view("<u4"), self.t + self.buf.shape[1], True)
        return h.astype("<u4").view(np.uint8).reshape(self.c, 32)
