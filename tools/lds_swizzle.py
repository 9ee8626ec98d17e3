#!/usr/bin/env python3
"""Finds, for every transform size, an XOR-linear LDS index swizzle that makes the row-NTT
kernel's LDS traffic bank-conflict free on gfx950, and prints the C++ table
(ligero_amd/csrc/lds_swizzle_table.h).

The kernel (ligero_amd/csrc/ntt_kernels.h) stores element `pos` of NTT slot `slot` at index
I = slot * K + pos in three planes (16 B, 16 B, 4 B per element).  The swizzle is
    sigma(I) = I ^ XOR_{j >= 3, bit j of I set} C[j]        with C[j] < 2^min(j, 5),
i.e. only the low five index bits are changed, as a GF(2)-linear function of the bits above
them AND of bits 3 and 4 (into the bits below them: the map is unit upper triangular), so sigma
is a bijection and sigma(a ^ b) = sigma(a) ^ sigma(b).  Feeding bits 3 and 4 in is what makes
the stride-8 accesses of the last pass conflict free: with inputs from bit 5 up only (round-1
first version) the best swizzles stayed at 1.5-1.7x the ideal cycle count; this family reaches
the ideal for every size.

Bank model (MI355X_MICROARCH.md, LDS table):
  ds_read_b128   groups {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63};
                 64 banks of 4 B: a lane occupies 4 banks, unit = (index mod 16)
  ds_write_b128  8 groups of 8 consecutive lanes, 32 banks: unit = (index mod 8)
  ds_read_b32 / ds_write_b32   2 groups of 32 lanes, 32 banks: bank = (index mod 32)
Lanes hitting the same bank with different addresses serialise.
"""
import itertools
import random
import sys

R128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
W128_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]
B32_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def plan(logk):
    first = logk if logk < 3 else (logk % 3 or 3)
    tpn = 1 if logk <= 3 else 1 << (logk - 3)
    wg = max(tpn, 256)
    return first, tpn, wg, wg // tpn


def accesses(logk):
    """yields (kind, [index per lane or None]) for every LDS wave-instruction of one workgroup's
    first wave(s); kind in {'r','w'} (each stands for b128+b128+b32 on the three planes)"""
    K = 1 << logk
    first, tpn, wg, npw = plan(logk)
    out = []
    passes = []
    logs = logk
    logr = first
    is_first = True
    while logs > 0:
        passes.append((logs, logr, is_first))
        logs -= logr
        logr = 3
        is_first = False
    nwaves = wg // 64
    for wave in range(min(nwaves, 2)):
        lanes = [wave * 64 + l for l in range(64)]
        for (logs, logr, is_first) in passes:
            logsub = logs - logr
            units = K >> logr
            it = 0
            while it * tpn < units:
                for q in range(1 << logr):
                    idx = []
                    for tid in lanes:
                        slot, t = divmod(tid, tpn)
                        u = t + it * tpn
                        if u >= units:
                            idx.append(None)
                            continue
                        blk, i0 = u >> logsub, u & ((1 << logsub) - 1)
                        idx.append(slot * K + (blk << logs) + i0 + (q << logsub))
                    if not is_first:
                        out.append(("r", idx))
                    out.append(("w", idx))
                it += 1
        # output stage: j = t, t + tpn, ... ; pos = digit reversal
        def dif_position(j):
            pos, logs, logr = 0, logk, first
            while logs > 0:
                logs -= logr
                pos += (j & ((1 << logr) - 1)) << logs
                j >>= logr
                logr = 3
            return pos
        j0 = 0
        while j0 < K:
            idx = []
            for tid in lanes:
                slot, t = divmod(tid, tpn)
                j = t + j0
                idx.append(slot * K + dif_position(j) if j < K else None)
            out.append(("r", idx))
            j0 += tpn
    return out


FIRST_BIT = 3


def sigma(i, C):
    x = i
    j = FIRST_BIT
    hi = i >> FIRST_BIT
    while hi:
        if hi & 1:
            x ^= C[j]
        hi >>= 1
        j += 1
    return x


def group_cycles(idx, groups, mod):
    cyc = 0
    for g in groups:
        banks = {}
        for l in g:
            if idx[l] is None:
                continue
            banks.setdefault(idx[l] % mod, set()).add(idx[l])
        cyc += max([len(v) for v in banks.values()], default=0)
    return cyc


def cost(acc, C):
    tot = ideal = 0
    for kind, idx in acc:
        s = [None if i is None else sigma(i, C) for i in idx]
        if kind == "r":
            tot += 2 * group_cycles(s, R128_GROUPS, 16) + group_cycles(s, B32_GROUPS, 32)
            ideal += 2 * 4 + 2
        else:
            tot += 2 * group_cycles(s, W128_GROUPS, 8) + group_cycles(s, B32_GROUPS, 32)
            ideal += 2 * 8 + 2
    return tot, ideal


def solve(logk, seed=0, iters=4000):
    K = 1 << logk
    _, _, _, npw = plan(logk)
    nbits = logk + (npw - 1).bit_length()
    acc = accesses(logk)
    rng = random.Random(seed)
    C = {j: 0 for j in range(FIRST_BIT, max(nbits, 6))}
    lim = {j: 1 << min(j, 5) for j in C}
    best, ideal = cost(acc, C)
    if best == ideal:
        return C, best, ideal
    # coordinate descent with random restarts over the 5-bit constants
    for restart in range(6):
        cur = dict(C) if restart == 0 else {j: rng.randrange(lim[j]) for j in C}
        cur_cost, _ = cost(acc, cur)
        improved = True
        while improved and cur_cost > ideal:
            improved = False
            for j in sorted(cur):
                bj, bc = cur[j], cur_cost
                for v in range(lim[j]):
                    if v == cur[j]:
                        continue
                    old = cur[j]
                    cur[j] = v
                    c, _ = cost(acc, cur)
                    if c < bc:
                        bj, bc = v, c
                    cur[j] = old
                if bc < cur_cost:
                    cur[j], cur_cost, improved = bj, bc, True
        if cur_cost < best:
            best, C = cur_cost, dict(cur)
        if best == ideal:
            break
    return C, best, ideal


def main():
    print("// generated by tools/lds_swizzle.py -- do not edit")
    print("// kLdsSwz[logk][j] = constant (< 2^min(j, 5)) XORed into the LDS index when bit j (j >= 3) of it is set")
    print("#pragma once")
    print("namespace lg {")
    print("constexpr unsigned char kLdsSwz[13][24] = {")
    print("    {0},")
    for logk in range(1, 13):
        C, best, ideal = solve(logk)
        base, _ = cost(accesses(logk), {j: 0 for j in C})
        row = [C.get(j, 0) for j in range(24)]
        print("    {%s},  // logk=%d: LDS cycles %d (ideal %d, unswizzled %d)" % (", ".join(str(v) for v in row), logk, best, ideal, base))
        sys.stdout.flush()
    print("};")
    print("}  // namespace lg")


if __name__ == "__main__":
    main()
