"""TEST INFRASTRUCTURE ONLY (oracle/): a Python restatement of the Fiat-Shamir pieces the reference's prover and verifier
draw on -- written from the crates' published algorithms, NOT from ligero_amd/host/transcript.hpp:

* ``ChaChaRng``            rand_chacha 0.3 ChaCha20Rng / ChaCha12Rng (`from_seed`, 64-bit block counter in words 12-13, stream id 0,
                           `next_u32` / `next_u64` off a 16-word block buffer) -- /root/reference/src/utils.rs:27, 36
* ``fr_rand``              ark-ff `UniformRand for Fp<MontBackend<_, 4>>`: four u64, the top limb masked to the modulus' bit length,
                           accepted when below the modulus, and kept AS the Montgomery limbs -- utils.rs:28
* ``gen_range``            rand 0.8 `UniformInt<usize>::sample_single` (widening multiply against a zone) -- utils.rs:47
* ``field_elements_from_seed`` / ``distinct_indices_from_seed``   utils.rs:23-29, 31-55
* ``PoseidonSponge``       ark-crypto-primitives `PoseidonSponge` with ark-poly-commit's `test_sponge()` parameters (rate 2, capacity 1,
                           8 full + 31 partial rounds, alpha 17, the fixed MDS, round constants from `test_rng()`), `absorb` of a
                           `Vec<u8>` and of a `Vec<F>`, `squeeze_bytes` -- /root/reference/src/ligero/tests.rs:151, 399 and the
                           call sites src/ligero/mod.rs:560, 653, 660, 719, 738, 839, 850, 941

PARITY UNPINNED against the Rust crates (un-vendored; the reference's tests hold no transcript bytes): pinned here are only the
RFC 8439 vector of the block function and agreement of independent restatements with each other."""
import struct

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R = 1 << 256
M32 = 0xFFFFFFFF


def chacha_block(rounds, key_words, w12_15):
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + list(w12_15)
    x = list(s)

    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & M32

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, s)]


class ChaChaRng:
    def __init__(self, seed: bytes, rounds: int):
        self.key = struct.unpack("<8I", seed)
        self.rounds, self.counter, self.buf = rounds, 0, []

    def next_u32(self):
        if not self.buf:
            self.buf = chacha_block(self.rounds, self.key, [self.counter & M32, self.counter >> 32, 0, 0])
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


def fr_rand(rng):
    """-> the element's Montgomery representation as an integer (limbs used as they are)"""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << 62) - 1
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < P:
            return v


def field_elements_from_seed(seed, n):
    rng = ChaChaRng(seed, 20)
    return [fr_rand(rng) for _ in range(n)]


def gen_range(rng, n):
    zone = ((n << (64 - n.bit_length())) - 1) & ((1 << 64) - 1)
    while True:
        m = rng.next_u64() * n
        if (m & ((1 << 64) - 1)) <= zone:
            return m >> 64


def distinct_indices_from_seed(seed, n, t):
    rng = ChaChaRng(seed, 20)
    sel = set()
    to_select = min(t, n - t)
    while len(sel) < to_select:
        sel.add(gen_range(rng, n))
    return sorted(sel) if to_select == t else [i for i in range(n) if i not in sel]


class PoseidonSponge:
    """canonical integers inside; rate 2, capacity 1"""
    RATE, CAP = 2, 1

    def __init__(self):
        rng = ChaChaRng(bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16), 12)
        rinv = pow(R, -1, P)
        self.full, self.partial, self.alpha = 8, 31, 17
        self.ark = [[fr_rand(rng) * rinv % P for _ in range(3)] for _ in range(39)]   # Montgomery limbs -> value
        self.mds = [[1, 0, 1], [1, 1, 0], [0, 1, 1]]
        self.state = [0, 0, 0]
        self.squeezing, self.idx = False, 0

    def permute(self):
        st = self.state
        for i in range(self.full + self.partial):
            st = [(a + b) % P for a, b in zip(st, self.ark[i])]
            if i < self.full // 2 or i >= self.full // 2 + self.partial:
                st = [pow(a, self.alpha, P) for a in st]
            else:
                st[0] = pow(st[0], self.alpha, P)
            st = [sum(a * b for a, b in zip(st, row)) % P for row in self.mds]
        self.state = st

    def _absorb(self, start, elems):
        while True:
            if start + len(elems) <= self.RATE:
                for i, e in enumerate(elems):
                    self.state[self.CAP + start + i] = (self.state[self.CAP + start + i] + e) % P
                self.squeezing, self.idx = False, start + len(elems)
                return
            take = self.RATE - start
            for i in range(take):
                self.state[self.CAP + start + i] = (self.state[self.CAP + start + i] + elems[i]) % P
            self.permute()
            elems, start = elems[take:], 0

    def absorb_elements(self, elems):
        if not elems:
            return
        if self.squeezing:
            self.permute()
            self._absorb(0, list(elems))
        else:
            idx = self.idx
            if idx == self.RATE:
                self.permute()
                idx = 0
            self._absorb(idx, list(elems))

    def absorb_bytes(self, data: bytes):
        b = len(data).to_bytes(8, "little") + data
        self.absorb_elements([int.from_bytes(b[i:i + 31], "little") for i in range(0, len(b), 31)])

    def _squeeze(self, start, n):
        out = []
        while True:
            left = n - len(out)
            if start + left <= self.RATE:
                out += self.state[self.CAP + start:self.CAP + start + left]
                self.squeezing, self.idx = True, start + left
                return out
            take = self.RATE - start
            out += self.state[self.CAP + start:self.CAP + start + take]
            if left != self.RATE:
                self.permute()
            start = 0

    def squeeze_elements(self, n):
        if not self.squeezing:
            self.permute()
            return self._squeeze(0, n)
        idx = self.idx
        if idx == self.RATE:
            self.permute()
            idx = 0
        return self._squeeze(idx, n)

    def squeeze_bytes(self, n):
        out = b"".join(e.to_bytes(32, "little")[:31] for e in self.squeeze_elements((n + 30) // 31))
        return out[:n]
