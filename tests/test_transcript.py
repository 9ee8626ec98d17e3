"""CPU tests of the Fiat-Shamir pieces (ligero_amd/host/transcript.hpp through libligero_host.so).
Pinned: the ChaCha20 block function (RFC 8439 section 2.3.2).  Everything else is checked against an
independent Python restatement (oracle/transcript_model.py) -- both are UNPINNED against the Rust
crates, so these tests catch implementation slips, not misreadings of the crates."""
import random

import numpy as np
import pytest

from oracle import transcript_model as tm
from ligero_amd import host_pipeline as hp

R = 1 << 256


def mont_ints(a):
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in np.asarray(a).reshape(-1, 4)]


def to_mont_limbs(vals):
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = v * R % tm.P
        for j in range(4):
            out[i, j] = (m >> (64 * j)) & ((1 << 64) - 1)
    return out


def test_chacha20_block_rfc8439():
    key = np.frombuffer(bytes(range(32)), dtype="<u4")
    got = hp.chacha_block(20, key, [1, 0x09000000, 0x4A000000, 0])
    want = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
            0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    assert [int(x) for x in got] == want
    assert tm.chacha_block(20, [int(k) for k in key], [1, 0x09000000, 0x4A000000, 0]) == want
    for rounds in (8, 12):
        assert [int(x) for x in hp.chacha_block(rounds, key, [7, 0, 0, 0])] == tm.chacha_block(rounds, [int(k) for k in key], [7, 0, 0, 0])


@pytest.mark.parametrize("seed", [bytes(32), bytes(range(32)), bytes([0xFF] * 32)])
def test_field_elements_and_indices_match_model(seed):
    got = mont_ints(hp.field_elements_from_seed(seed, 300))
    assert got == tm.field_elements_from_seed(seed, 300)
    assert all(v < tm.P for v in got)
    for n, t in ((1024, 156), (32, 32), (32768, 156), (64, 40), (8, 0)):
        idx = [int(x) for x in hp.distinct_indices_from_seed(seed, n, t)]
        assert idx == tm.distinct_indices_from_seed(seed, n, t)
        assert len(idx) == t and idx == sorted(set(idx)) and all(0 <= i < n for i in idx)


def test_poseidon_sponge_matches_model():
    rng = random.Random(11)
    a, b = hp.PoseidonSponge(), tm.PoseidonSponge()
    for step in range(60):
        op = rng.choice(("bytes", "elems", "sq_bytes", "sq_elems"))
        if op == "bytes":
            data = bytes(rng.randrange(256) for _ in range(rng.choice((0, 1, 31, 32, 33, 70))))
            a.absorb_bytes(data)
            b.absorb_bytes(data)
        elif op == "elems":
            vals = [rng.randrange(tm.P) for _ in range(rng.choice((0, 1, 2, 3, 5, 128)))]
            a.absorb_elements(to_mont_limbs(vals))
            b.absorb_elements(vals)
        elif op == "sq_bytes":
            n = rng.choice((1, 31, 32, 62, 63, 100))
            assert a.squeeze_bytes(n) == b.squeeze_bytes(n), step
        else:
            n = rng.choice((1, 2, 3, 4, 7))
            got = [v * pow(R, -1, tm.P) % tm.P for v in mont_ints(a.squeeze_elements(n))]
            assert got == b.squeeze_elements(n), step


def test_transcript_order_of_prove_is_reproducible():
    """two sponges fed the same absorbs give the same seeds (what keeps prover and verifier in step)"""
    outs = []
    for _ in range(2):
        s = hp.PoseidonSponge()
        s.absorb_bytes(bytes(range(32)))
        seed1 = s.squeeze_bytes(32)
        s.absorb_elements(hp.field_elements_from_seed(seed1, 5))
        outs.append((seed1, s.squeeze_bytes(32)))
    assert outs[0] == outs[1] and outs[0][0] != outs[0][1]


def test_eight_sponges_in_lock_step_equal_eight_separate_sponges():
    """absorb_elements_x8 (AVX-512 IFMA where the host has it: eight sponges on the lanes of one vector, radix-2^52 Montgomery
    products; eight ordinary calls elsewhere, or when the sponges fall out of step) leaves the same states as the scalar sponge:
    random lengths, field corners, mode switches in between, sponges deliberately out of step"""
    rng = random.Random(23)
    a = [hp.PoseidonSponge() for _ in range(8)]
    b = [hp.PoseidonSponge() for _ in range(8)]
    print("IFMA path available:", hp.ifma_available())
    for step in range(14):
        count = (1, 2, 3, 0)[step] if step < 4 else rng.randrange(1, 260)
        vals = [[rng.randrange(tm.P) for _ in range(count)] for _ in range(8)]
        if step == 5:
            vals[2][0], vals[3][0], vals[4][0] = 0, tm.P - 1, 1
        limbs = np.stack([to_mont_limbs(v) if count else np.zeros((0, 4), dtype=np.uint64) for v in vals]) if count else np.zeros((8, 0, 4), dtype=np.uint64)
        for j in range(8):
            a[j].absorb_elements(limbs[j])
        hp.sponges_absorb_elements_x8(b, limbs)
        if step % 3 == 2:                                       # squeeze: the next absorb starts with a permutation
            n = rng.choice((1, 32, 63))
            for j in range(8):
                assert a[j].squeeze_bytes(n) == b[j].squeeze_bytes(n), (step, j)
        if step == 7:                                           # sponge 6 falls out of step: the group falls back, still equal
            a[6].absorb_bytes(b"x")
            b[6].absorb_bytes(b"x")
        if step == 9:                                           # back in step after everybody squeezed
            for j in range(8):
                assert a[j].squeeze_bytes(32) == b[j].squeeze_bytes(32)
    for j in range(8):
        assert a[j].squeeze_bytes(64) == b[j].squeeze_bytes(64), j
