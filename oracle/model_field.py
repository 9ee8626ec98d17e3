"""Python big-int model of the hot path over ANY prime field with a radix-2 domain -- the reference is generic over
`F: PrimeField` (src/ligero/mod.rs:146) and its tests instantiate ark_bn254::Fr and ark_bls12_377::Fq (tests.rs:23-24).

TEST INFRASTRUCTURE ONLY (see oracle/model.py, whose Merkle functions this reuses).  Same statements as model.py with the
modulus, the 2-adic root and the canonical byte length of an element as parameters:
  reed_solomon_interpolate / evaluate   mod.rs:998-1008   (GeneralEvaluationDomain = Radix2 domain, group_gen = root^(2^s / size))
  col_hash                              mod.rs:536-542, types.rs:18: Blake2s-256(LE64(len) || elements as ceil(bits / 8)-byte... 
                                        CanonicalSerialize of Fp<_, N> writes N * 8 bytes little endian (48 for the 6-limb Fq)
PARITY UNPINNED like model.py; additionally restated from memory: ark_bls12_377::Fq's GENERATOR = 15 (fixes which primitive
2^46-th root the domains use)."""
from __future__ import annotations

import hashlib
import struct
from typing import List, Sequence

from . import model


class Field:
    def __init__(self, name: str, p: int, generator: int, limbs64: int):
        self.name, self.p, self.generator, self.limbs = name, p, generator, limbs64
        self.nbytes = 8 * limbs64
        t, s = p - 1, 0
        while t % 2 == 0:
            t //= 2
            s += 1
        self.two_adicity = s
        self.root = pow(generator, (p - 1) >> s, p)
        assert pow(self.root, 1 << (s - 1), p) == p - 1, "generator is not a quadratic non-residue"
        self.R = (1 << (64 * limbs64)) % p

    def domain_generator(self, size: int) -> int:
        assert size & (size - 1) == 0 and 1 <= size <= (1 << self.two_adicity)
        return pow(self.root, (1 << self.two_adicity) // size, self.p)

    def ntt(self, coeffs: Sequence[int], omega: int) -> List[int]:
        p = self.p
        a = list(coeffs)
        n = len(a)
        model._bitrev_permute(a)
        length = 2
        while length <= n:
            wl = pow(omega, n // length, p)
            half = length >> 1
            for start in range(0, n, length):
                w = 1
                for i in range(start, start + half):
                    u, v = a[i], a[i + half] * w % p
                    a[i], a[i + half] = (u + v) % p, (u - v) % p
                    w = w * wl % p
            length <<= 1
        return a

    def reed_solomon_interpolate(self, msg: Sequence[int], k: int) -> List[int]:
        m = list(msg) + [0] * (k - len(msg))
        kinv = pow(k, -1, self.p)
        return [x * kinv % self.p for x in self.ntt(m, pow(self.domain_generator(k), -1, self.p))]

    def reed_solomon_evaluate(self, coeffs: Sequence[int], n: int) -> List[int]:
        return self.ntt(list(coeffs) + [0] * (n - len(coeffs)), self.domain_generator(n))

    def col_hash(self, col: Sequence[int]) -> bytes:
        h = hashlib.blake2s(digest_size=32)
        h.update(struct.pack("<Q", len(col)))
        for x in col:
            h.update(x.to_bytes(self.nbytes, "little"))
        return h.digest()

    def encode_commit(self, preenc_u: Sequence[Sequence[int]], k: int, n: int):
        coeffs = [self.reed_solomon_interpolate(row, k) for row in preenc_u]
        u = [self.reed_solomon_evaluate(c, n) for c in coeffs]
        leaves = [self.col_hash([row[j] for row in u]) for j in range(n)]
        nodes = model.merkle_tree(leaves)
        return coeffs, u, leaves, nodes, nodes[0]

    # conversions to / from the ABI's Montgomery limbs (numpy uint64 (..., limbs))
    def to_mont_limbs(self, ints):
        import numpy as np
        out = np.empty((len(ints), self.limbs), dtype=np.uint64)
        mask = (1 << 64) - 1
        for i, v in enumerate(ints):
            m = v * self.R % self.p
            out[i] = [(m >> (64 * l)) & mask for l in range(self.limbs)]
        return out

    def from_mont_limbs(self, arr) -> List[int]:
        rinv = pow(self.R, -1, self.p)
        flat = arr.reshape(-1, self.limbs)
        return [sum(int(x[l]) << (64 * l) for l in range(self.limbs)) * rinv % self.p for x in flat]


BN254_FR = Field("ark_bn254::Fr", model.P, 5, 4)
BLS12_377_FQ = Field("ark_bls12_377::Fq",
                     258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177, 15, 6)
