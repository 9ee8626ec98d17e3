#!/usr/bin/env python3
"""vectors for / check of tools/microbench4.hip's shoup29 leg:  mb4.py gen | mb4.py check"""
import random
import struct
import sys

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
M = (1 << 29) - 1


def limbs(x):
    return [(x >> (29 * i)) & M for i in range(8)] + [x >> 232]


def dirty(rng):
    # limbs 0..7 up to 6*2^29, value < 2^261
    while True:
        v = [rng.randrange(0, 6 << 29) for _ in range(8)] + [rng.randrange(0, 1 << 28)]
        if sum(l << (29 * i) for i, l in enumerate(v)) < (1 << 261):
            return v


def gen(n=4096):
    rng = random.Random(7)
    out = []
    for t in range(n):
        if t % 4 == 0:
            a = limbs(rng.randrange(0, 1 << 261))
        elif t % 4 == 1:
            a = dirty(rng)
        elif t % 4 == 2:
            a = [6 << 29] * 8 + [(1 << 28) - 100]          # the largest dirty operand
        else:
            a = limbs(rng.choice([0, 1, P - 1, P, 2 * P, (1 << 261) - 1]))
        w = rng.choice([0, 1, P - 1, rng.randrange(P), rng.randrange(P)])
        wq = (w << 261) // P
        out += a + limbs(w) + limbs(wq)
    open("tools/mb4_vec.bin", "wb").write(struct.pack(f"<{len(out)}I", *out))


def check():
    vin = open("tools/mb4_vec.bin", "rb").read()
    vout = open("tools/mb4_out.bin", "rb").read()
    n = len(vin) // (27 * 4)
    a = struct.unpack(f"<{n * 27}I", vin)
    r = struct.unpack(f"<{n * 9}I", vout)
    worst = 0
    for t in range(n):
        val = lambda v: sum(l << (29 * i) for i, l in enumerate(v))
        av, wv = val(a[t * 27:t * 27 + 9]), val(a[t * 27 + 9:t * 27 + 18])
        rl = r[t * 9:t * 9 + 9]
        rv = val(rl)
        assert all(l <= M for l in rl[:8]), (t, rl)
        assert rv % P == av * wv % P, (t, "value")
        assert rv < 3 * P, (t, rv / P)
        worst = max(worst, rv // P)
    print(f"shoup29: {n} vectors ok, result < {worst + 1} p")


if __name__ == "__main__":
    gen() if sys.argv[1] == "gen" else check()
