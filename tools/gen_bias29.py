#!/usr/bin/env python3
"""Prints the Bias29<K, L> tables of ligero_amd/csrc/fr29_gfx950.h: K*p in radix 2^29 with limbs
0..7 raised by 2^L (limb i >= 1 and limb 8 pay back 2^(L-29)), so that a - b + bias is limb-wise
non-negative for any b with limbs 0..7 <= 2^L - 2^(L-29) and value(b) < (K - 1) p."""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
M = (1 << 29) - 1


def limbs29(x):
    return [(x >> (29 * i)) & M for i in range(8)] + [x >> 232]


def bias(K, L):
    v = limbs29(K * P)
    m = [v[0] + (1 << L)] + [v[i] + (1 << L) - (1 << (L - 29)) for i in range(1, 8)] + [v[8] - (1 << (L - 29))]
    assert sum(x << (29 * i) for i, x in enumerate(m)) == K * P
    assert all(0 <= x < 2**32 for x in m)
    return m


if __name__ == "__main__":
    for K, L in ((4, 29), (8, 30), (16, 30)):
        print(f"Bias29<{K}, {L}>:", ", ".join("0x%08xu" % x for x in bias(K, L)))
    print("one261 =", ", ".join("0x%08xu" % x for x in limbs29((1 << 261) % P)))
