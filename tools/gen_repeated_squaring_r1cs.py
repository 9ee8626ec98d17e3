#!/usr/bin/env python3
"""Synthetic repeated-squaring R1CS + witness for BASELINE.json configs[2] / [3]:

    N constraints  w_{i+1} = w_i * w_i   (A = [(1, w_i)], B = [(1, w_i)], C = [(1, w_{i+1})]),
    wires [1, w_0, ..., w_N], w_0 derived from the seed (SplitMix64 output reduced mod r), no empty rows

-- the shape of the reference's circom/repeated_squaring_10.circom (source only there: no .r1cs is shipped) scaled to
N = 2^20 / 2^22, written as a circom .r1cs v1 file ("r1cs", version 1, sections header / constraints /
wire map) and a snarkjs .wtns witness, i.e. exactly what ArithmeticCircuit::from_constraint_system
(src/arithmetic_circuit/mod.rs:455-520) and the reference's witness loader consume for the circom fixtures.

    python tools/gen_repeated_squaring_r1cs.py <log2 N> <seed> <out.r1cs> <out.wtns>

Expected LigeroCircuit dimensions (src/ligero/mod.rs:171-175, 275-294), asserted by the tests:
    N = 2^20: 5 242 883 nodes, sol_vec_length 6 291 458, (m, k, n, t) = (2509, 4096, 32 768, 156)
    N = 2^22: 20 971 523 nodes, (m, k, n, t) = (5017, 8192, 65 536, 156)
"""
import struct
import sys

import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def splitmix64(seed: int) -> int:
    z = (seed + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    return z ^ (z >> 31)


def witness(log_n: int, seed: int):
    """[1, w_0, ..., w_N] as python ints"""
    n = 1 << log_n
    w = splitmix64(seed) % P
    if w < 2:
        w += 2
    out = [1, w]
    for _ in range(n):
        w = w * w % P
        out.append(w)
    return out


def write_r1cs(path: str, log_n: int):
    n = 1 << log_n
    n_wires = n + 2
    prime = P.to_bytes(32, "little")
    header = struct.pack("<I", 32) + prime + struct.pack("<IIIIQI", n_wires, 1, 0, 1, n_wires, n)   # 1 public output (w_N), 1 private input (w_0)
    # one constraint = three linear combinations of one term each: (u32 nnz = 1, u32 wire, 32-byte LE coefficient = 1)
    lc = np.dtype([("nnz", "<u4"), ("wire", "<u4"), ("coef", "u1", 32)])
    cons = np.zeros((n, 3), dtype=lc)
    cons["nnz"] = 1
    cons["coef"][:, :, 0] = 1
    i = np.arange(n, dtype=np.uint32)
    cons["wire"][:, 0] = 1 + i
    cons["wire"][:, 1] = 1 + i
    cons["wire"][:, 2] = 2 + i
    body = cons.tobytes()
    wiremap = np.arange(n_wires, dtype="<u8").tobytes()
    with open(path, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3))
        f.write(struct.pack("<IQ", 1, len(header)) + header)
        f.write(struct.pack("<IQ", 2, len(body)))
        f.write(body)
        f.write(struct.pack("<IQ", 3, len(wiremap)) + wiremap)


def write_wtns(path: str, values):
    prime = P.to_bytes(32, "little")
    sec1 = struct.pack("<I", 32) + prime + struct.pack("<I", len(values))
    with open(path, "wb") as f:
        f.write(b"wtns" + struct.pack("<II", 2, 2))
        f.write(struct.pack("<IQ", 1, len(sec1)) + sec1)
        f.write(struct.pack("<IQ", 2, 32 * len(values)))
        f.write(b"".join(v.to_bytes(32, "little") for v in values))


def main():
    if len(sys.argv) != 5:
        sys.exit(__doc__)
    log_n, seed = int(sys.argv[1]), int(sys.argv[2])
    write_r1cs(sys.argv[3], log_n)
    write_wtns(sys.argv[4], witness(log_n, seed))


if __name__ == "__main__":
    main()
