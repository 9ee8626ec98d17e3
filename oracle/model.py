"""Python big-int model of the Ligero encode-and-commit hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: it
may be imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker.

PARITY UNPINNED ("Oracle"): the reference (NP-Eng/ligero, Rust)
cannot be built here (no cargo/rustc, arkworks crates are not vendored) and its
tests hold no golden bytes for this path.  What *is* pinned:

* Reed-Solomon interpolate/evaluate are mathematically determined by the
  domain generators; the 2-adic root below is recomputed from the modulus read
  out of the reference's own ``.r1cs`` fixtures and generator 5.
* Blake2s-256 / SHA-256 are checked against ``hashlib`` and RFC vectors.
* The R1CS -> circuit -> (x, y, z, w) restatement reproduces the node counts the
  reference's tests assert (cube: 15 nodes, arithmetic_circuit/tests.rs:239)
  and the reference's own acceptance condition (all outputs evaluate to 1,
  ligero/tests.rs:391-394) on the reference's Poseidon fixture.

What is restated from the published arkworks 0.5 / ark-poly-commit
(HungryCatsStudio/poly-commit @ release-0.5) algorithms and NOT validated
against a run of the reference: the byte framing of the column-hash input
(u64-LE length prefix), the ``LE64(32)`` prefixes in the bottom Merkle level,
and heap / root->leaf ordering of authentication paths.

Each function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import hashlib
import json
import math
import struct
from typing import List, Sequence, Tuple

# --------------------------------------------------------------------------
# BN254 Fr (ark_bn254::Fr; reference uses it at src/ligero/tests.rs:24)
# --------------------------------------------------------------------------
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R = (1 << 256) % P                      # Montgomery radix (ark-ff MontBackend, 4 x u64)
R2 = (R * R) % P
RINV = pow(R, -1, P)
INV64 = (-pow(P, -1, 1 << 64)) % (1 << 64)
TWO_ADICITY = 28
GENERATOR = 5
TWO_ADIC_ROOT = pow(GENERATOR, (P - 1) >> TWO_ADICITY, P)

assert INV64 == 0xC2E1F593EFFFFFFF
assert pow(TWO_ADIC_ROOT, 1 << TWO_ADICITY, P) == 1
assert pow(TWO_ADIC_ROOT, 1 << (TWO_ADICITY - 1), P) == P - 1


def to_mont(a: int) -> int:
    return (a * R) % P


def from_mont(a: int) -> int:
    return (a * RINV) % P


def domain_generator(size: int) -> int:
    """GeneralEvaluationDomain::new(size).group_gen for a power-of-two size
    (reference call sites: src/ligero/mod.rs:204-212)."""
    assert size & (size - 1) == 0 and 1 <= size <= (1 << TWO_ADICITY)
    return pow(TWO_ADIC_ROOT, (1 << TWO_ADICITY) // size, P)


# --------------------------------------------------------------------------
# NTT (values are plain integers mod P here)
# --------------------------------------------------------------------------
def _bitrev_permute(a: List[int]) -> None:
    n = len(a)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]


def ntt(coeffs: Sequence[int], omega: int) -> List[int]:
    """out[j] = sum_d coeffs[d] * omega^(j d): natural order in and out
    (EvaluationDomain::fft semantics, reference call site mod.rs:1007)."""
    a = list(coeffs)
    n = len(a)
    assert n & (n - 1) == 0
    _bitrev_permute(a)
    length = 2
    while length <= n:
        wl = pow(omega, n // length, P)
        half = length >> 1
        for start in range(0, n, length):
            w = 1
            for i in range(start, start + half):
                u = a[i]
                v = a[i + half] * w % P
                a[i] = (u + v) % P
                a[i + half] = (u - v) % P
                w = w * wl % P
        length <<= 1
    return a


def intt(evals: Sequence[int], omega: int) -> List[int]:
    """EvaluationDomain::ifft (reference call site mod.rs:1001)."""
    n = len(evals)
    ninv = pow(n, -1, P)
    return [x * ninv % P for x in ntt(evals, pow(omega, -1, P))]


def naive_dft(coeffs: Sequence[int], omega: int) -> List[int]:
    n = len(coeffs)
    return [sum(c * pow(omega, j * d, P) for d, c in enumerate(coeffs)) % P for j in range(n)]


# --------------------------------------------------------------------------
# Reed-Solomon (src/ligero/mod.rs:998-1012)
# --------------------------------------------------------------------------
def reed_solomon_interpolate(msg: Sequence[int], k: int) -> List[int]:
    """mod.rs:998-1002: resize to k, small_domain.ifft."""
    m = list(msg) + [0] * (k - len(msg))
    return intt(m, domain_generator(k))


def reed_solomon_evaluate(coeffs: Sequence[int], n: int) -> List[int]:
    """mod.rs:1004-1008: resize to n, large_domain.fft."""
    c = list(coeffs) + [0] * (n - len(coeffs))
    return ntt(c, domain_generator(n))


def reed_solomon(msg: Sequence[int], k: int, n: int) -> List[int]:
    """mod.rs:1010-1012."""
    return reed_solomon_evaluate(reed_solomon_interpolate(msg, k), n)


# --------------------------------------------------------------------------
# Column hash and Merkle tree (mod.rs:536-551; src/ligero/types.rs:15-46)
# --------------------------------------------------------------------------
def fr_to_bytes(a: int) -> bytes:
    """CanonicalSerialize of one Fr: 32 bytes LE of the canonical integer."""
    return a.to_bytes(32, "little")


def col_hash(col: Sequence[int]) -> bytes:
    """FieldToBytesColHasher<F, Blake2s256>::evaluate (types.rs:18):
    Blake2s-256( serialize_compressed(Vec<F>) ) = LE64(len) || elements."""
    h = hashlib.blake2s(digest_size=32)
    h.update(struct.pack("<Q", len(col)))
    for x in col:
        h.update(fr_to_bytes(x))
    return h.digest()


def _leaf_convert(d: bytes) -> bytes:
    """ByteDigestConverter<Vec<u8>>: uncompressed serialization of the leaf
    digest = LE64(len) || bytes (TestMerkleTreeParams, types.rs:6-8)."""
    return struct.pack("<Q", len(d)) + d


def merkle_tree(leaves: Sequence[bytes]) -> List[bytes]:
    """create_merkle_tree (mod.rs:544-549) with LeafIdentityHasher + Sha256:
    returns the n-1 inner nodes in heap order (root = index 0).  Leaves are
    padded to a power of two with the default (empty) leaf, as upstream does;
    never triggers in Ligero since n = 8k."""
    lv = list(leaves)
    n = 1
    while n < len(lv):
        n <<= 1
    lv += [b""] * (n - len(lv))
    assert n >= 2
    nodes: List[bytes] = [b""] * (n - 1)
    base = n // 2 - 1                      # first index of the bottom inner level
    for i in range(n // 2):
        nodes[base + i] = hashlib.sha256(_leaf_convert(lv[2 * i]) + _leaf_convert(lv[2 * i + 1])).digest()
    for i in range(base - 1, -1, -1):
        nodes[i] = hashlib.sha256(nodes[2 * i + 1] + nodes[2 * i + 2]).digest()
    return nodes


def merkle_path(leaves: Sequence[bytes], nodes: Sequence[bytes], index: int) -> Tuple[bytes, List[bytes]]:
    """MerkleTree::generate_proof (call site mod.rs:951): (leaf_sibling_hash,
    auth_path ordered root-side first, length log2(n) - 1)."""
    n = len(leaves)
    sib = leaves[index ^ 1]
    cur = (n // 2 - 1) + (index >> 1)      # parent of the leaf, heap index
    path = []
    while cur != 0:
        s = cur + 1 if cur & 1 else cur - 1
        path.append(nodes[s])
        cur = (cur - 1) >> 1
    path.reverse()
    return sib, path


def merkle_verify(root: bytes, leaf: bytes, index: int, sib: bytes, path: Sequence[bytes]) -> bool:
    """Path::verify (call site mod.rs:985-995)."""
    l, r = (leaf, sib) if index & 1 == 0 else (sib, leaf)
    cur = hashlib.sha256(_leaf_convert(l) + _leaf_convert(r)).digest()
    idx = index >> 1
    for s in reversed(path):
        cur = hashlib.sha256((cur + s) if idx & 1 == 0 else (s + cur)).digest()
        idx >>= 1
    return cur == root


# --------------------------------------------------------------------------
# The hot path: mod.rs:521-551 and the openings mod.rs:935-955
# --------------------------------------------------------------------------
def encode_commit(preenc_u: Sequence[Sequence[int]], k: int, n: int):
    """preenc_u: rows x k plain integers.  Returns (coeffs, U, leaves, nodes, root)."""
    coeffs = [reed_solomon_interpolate(row, k) for row in preenc_u]        # mod.rs:521-526
    u = [reed_solomon_evaluate(c, n) for c in coeffs]                       # mod.rs:528-533
    columns = [[row[j] for row in u] for j in range(n)]                     # matrices/mod.rs:163-167
    leaves = [col_hash(c) for c in columns]                                 # mod.rs:536-542
    nodes = merkle_tree(leaves)                                             # mod.rs:544-549
    return coeffs, u, leaves, nodes, nodes[0]                               # mod.rs:551


def open_columns(u, leaves, nodes, indices):
    """mod.rs:944-952 (the index derivation at 941-942 is Fiat-Shamir, host side)."""
    cols = [[row[i] for row in u] for i in indices]                         # matrices/mod.rs:169-171
    paths = [merkle_path(leaves, nodes, i) for i in indices]
    return cols, paths


# --------------------------------------------------------------------------
# Dimensions (mod.rs:171-175, 275-294) and calculate_t (ark-poly-commit
# linear_codes/utils.rs, restated)
# --------------------------------------------------------------------------
def compute_dimensions(sol_vec_length: int) -> Tuple[int, int]:
    m = math.ceil(math.sqrt(float(sol_vec_length)))
    k = 1
    while k < m:
        k <<= 1
    return m, k


def calculate_t(sec_param: int, distance: Tuple[int, int], codeword_len: int, field_bits: int = 254) -> int:
    residual = codeword_len / 2.0 ** field_bits
    rhs = math.log2(2.0 ** (-sec_param) - residual)
    nom = rhs - 1.0
    denom = math.log2(1.0 - 0.5 * distance[0] / distance[1])
    t = math.ceil(nom / denom)
    return t if t < codeword_len else codeword_len


def reed_solomon_parameters(m: int, k: int, lam: int) -> Tuple[int, int]:
    n = 8 * k
    return n, calculate_t(lam, (n - k + 1, n), n)


# --------------------------------------------------------------------------
# Input side (only to GENERATE preenc_u from the reference's fixtures):
# .r1cs v1 reader, from_constraint_system, evaluation trace, x/y/z/w assembly
# --------------------------------------------------------------------------
def read_r1cs(path: str):
    """circom .r1cs v1.  Returns (prime, n_wires, constraints)
    with constraints = [(A, B, C)], each a list of (coeff, wire)."""
    data = open(path, "rb").read()
    assert data[:4] == b"r1cs"
    version, nsec = struct.unpack_from("<II", data, 4)
    assert version == 1
    off = 12
    sections = {}
    for _ in range(nsec):
        typ, length = struct.unpack_from("<IQ", data, off)
        off += 12
        sections[typ] = (off, length)
        off += length
    o, _ = sections[1]
    fs = struct.unpack_from("<I", data, o)[0]
    prime = int.from_bytes(data[o + 4:o + 4 + fs], "little")
    n_wires, n_pub_out, n_pub_in, n_prv_in = struct.unpack_from("<IIII", data, o + 4 + fs)
    n_labels, n_constraints = struct.unpack_from("<QI", data, o + 4 + fs + 16)
    o, _ = sections[2]
    constraints = []
    for _ in range(n_constraints):
        lcs = []
        for _ in range(3):
            nnz = struct.unpack_from("<I", data, o)[0]
            o += 4
            lc = []
            for _ in range(nnz):
                wire = struct.unpack_from("<I", data, o)[0]
                coeff = int.from_bytes(data[o + 4:o + 4 + fs], "little")
                o += 4 + fs
                lc.append((coeff, wire))
            lcs.append(lc)
        constraints.append(tuple(lcs))
    return prime, n_wires, constraints


class ArithmeticCircuit:
    """Restatement of src/arithmetic_circuit/mod.rs:27-244 (only what
    from_constraint_system uses)."""

    def __init__(self):
        self.nodes: List[tuple] = []        # ('V', label), ('C', v), ('A', l, r), ('M', l, r)
        self.constants = {}
        self.variables = {}                 # label -> node index (mod.rs:33-34)

    def constant(self, v: int) -> int:      # mod.rs:76-84
        v %= P
        if v in self.constants:
            return self.constants[v]
        self.nodes.append(("C", v))
        self.constants[v] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def new_variable_with_label(self, label: str) -> int:   # mod.rs:92-100
        self.nodes.append(("V", label))
        if label in self.variables:
            raise ValueError(f"Variable label already in use: {label}")
        self.variables[label] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def new_variable(self) -> int:          # mod.rs:107-109
        return self.new_variable_with_label(f"var_{len(self.variables)}")

    def new_variables(self, num: int) -> List[int]:   # mod.rs:111-113
        return [self.new_variable() for _ in range(num)]

    def last(self) -> int:                  # mod.rs:51-53
        return len(self.nodes) - 1

    def add(self, l: int, r: int) -> int:   # mod.rs:125-131
        self.nodes.append(("A", l, r))
        return len(self.nodes) - 1

    def mul(self, l: int, r: int) -> int:   # mod.rs:139-145
        self.nodes.append(("M", l, r))
        return len(self.nodes) - 1

    def add_nodes(self, idx: Sequence[int]) -> int:   # mod.rs:148-153
        acc = idx[0]
        for i in idx[1:]:
            acc = self.add(acc, i)
        return acc

    def mul_nodes(self, idx: Sequence[int]) -> int:   # mod.rs:156-161
        acc = idx[0]
        for i in idx[1:]:
            acc = self.mul(acc, i)
        return acc

    def pow(self, node: int, exponent: int) -> int:   # mod.rs:164-205: square-and-multiply over the bits after the leading one
        cur = node
        for bit in bin(exponent)[3:]:
            cur = self.mul(cur, cur)
            if bit == "1":
                cur = self.mul(cur, node)
        return cur

    def minus(self, node: int) -> int:      # mod.rs:225-228
        return self.mul(self.constant(P - 1), node)

    def compile_sparse_scalar_product(self, row) -> int:   # mod.rs:501-520
        consts = [(self.constant(c), w) for c, w in row]
        prods = [(ci + w) if (ci == 0 or w == 0) else self.mul(ci, w) for ci, w in consts]
        return self.add_nodes(prods)


def from_constraint_system(n_wires: int, constraints):
    """src/arithmetic_circuit/mod.rs:455-495.  Zero coefficients are dropped as
    ark-relations' to_matrices does."""
    c = ArithmeticCircuit()
    one = c.constant(1)
    for _ in range(n_wires - 1):
        c.new_variable()
    strip = lambda lc: [(v % P, w) for v, w in lc if v % P != 0]
    a = [c.compile_sparse_scalar_product(strip(A)) for A, _, _ in constraints]
    b = [c.compile_sparse_scalar_product(strip(B)) for _, B, _ in constraints]
    cc = [c.compile_sparse_scalar_product(strip(C)) for _, _, C in constraints]
    ab = [c.mul(x, y) for x, y in zip(a, b)]
    minus_one = c.constant(P - 1)
    minus_c = [c.mul(x, minus_one) for x in cc]
    outputs = [c.add_nodes([x, y, one]) for x, y in zip(ab, minus_c)]
    return c, outputs


def evaluation_trace(circ: ArithmeticCircuit, assignment: Sequence[Tuple[int, int]]) -> List[int]:
    """mod.rs:325-358.  Nodes only reference earlier nodes, so a forward sweep
    computes the same values as the reference's recursive evaluator; nodes the
    outputs do not depend on would be None upstream (and make prove panic,
    src/ligero/mod.rs:477) -- callers of this model use circuits where every
    node is reachable."""
    vals: List[int] = [0] * len(circ.nodes)
    given = dict(assignment)
    for i, nd in enumerate(circ.nodes):
        if nd[0] == "C":
            vals[i] = nd[1]
        elif nd[0] == "V":
            vals[i] = given[i] % P
        elif nd[0] == "A":
            vals[i] = (vals[nd[1]] + vals[nd[2]]) % P
        else:
            vals[i] = vals[nd[1]] * vals[nd[2]] % P
    return vals


def ligero_dims(circ: ArithmeticCircuit, n_outputs: int, lam: int = 128):
    """src/ligero/mod.rs:171-175."""
    sol_vec_length = 1 + len(circ.nodes) - len(circ.constants) + n_outputs
    m, k = compute_dimensions(sol_vec_length)
    n, t = reed_solomon_parameters(m, k, lam)
    return m, k, n, t


def build_preenc_u(circ: ArithmeticCircuit, sol: Sequence[int], m: int, k: int) -> List[List[int]]:
    """src/ligero/mod.rs:483-516: x, y, z, w -> [X; Y; Z; W] (4m x k)."""
    x, y, z, w = [], [], [], []
    for i, (val, nd) in enumerate(zip(sol, circ.nodes)):
        if nd[0] == "C" and i != 0:
            continue
        w.append(val)
        if nd[0] == "M":
            x.append(sol[nd[1]]); y.append(sol[nd[2]]); z.append(val)
        else:
            x.append(0); y.append(0); z.append(0)
    out = []
    for vec in (x, y, z, w):
        assert len(vec) <= m * k
        vec = vec + [0] * (m * k - len(vec))
        out += [vec[i * k:(i + 1) * k] for i in range(m)]          # as_matrix, mod.rs:1014-1017
    return out


def preenc_from_r1cs(r1cs_path: str, witness: Sequence[int], lam: int = 128):
    """R1CS + full witness (wire 0 = 1) -> (m, k, n, t, preenc_u).  The constant 1
    is node 0 already (from_constraint_system puts it there), so
    LigeroCircuit::new's insert_one / bump_index (mod.rs:160-169) are identities."""
    prime, n_wires, cons = read_r1cs(r1cs_path)
    assert prime == P and len(witness) == n_wires
    circ, outputs = from_constraint_system(n_wires, cons)
    sol = evaluation_trace(circ, [(i, v) for i, v in enumerate(witness) if i >= 1])
    assert all(sol[o] == 1 for o in outputs), "witness does not satisfy the R1CS"
    m, k, n, t = ligero_dims(circ, len(outputs), lam)
    return m, k, n, t, build_preenc_u(circ, sol, m, k), circ, outputs


def load_witness_json(path: str) -> List[int]:
    return [int(s) for s in json.load(open(path))]


# --------------------------------------------------------------------------
# Synthetic inputs shared by tests / bench (seeded, data-independent cost)
# --------------------------------------------------------------------------
def splitmix64(seed: int):
    s = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        s = (s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def random_elements(seed: int, count: int) -> List[int]:
    g = splitmix64(seed)
    out = []
    for _ in range(count):
        v = next(g) | (next(g) << 64) | (next(g) << 128) | (next(g) << 192)
        out.append(v % P)
    return out


# --------------------------------------------------------------------------
# The three sub-proof polynomials: the device-friendly arithmetic of
# prove_interleaved / prove_linear_constraints / prove_quadratic_constraints.  Challenges
# (r vectors) are inputs: deriving them is Fiat-Shamir, host side.
# --------------------------------------------------------------------------
def poly_mul(a: Sequence[int], b: Sequence[int]) -> List[int]:
    """DensePolynomial * DensePolynomial (exact; any algorithm gives these coefficients)"""
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % P
    return out


def dense_row_mul(rows: Sequence[Sequence[int]], r: Sequence[int]) -> List[int]:
    """DenseMatrix::row_mul (src/matrices/mod.rs:138-149): result[c] = sum_i rows[i][c] * r[i];
    prove_interleaved calls it as preenc_u.row_mul(&r_interleaved) (src/ligero/mod.rs:658)"""
    out = [0] * len(rows[0])
    for c, row in zip(r, rows):
        for j, v in enumerate(row):
            out[j] = (out[j] + v * c) % P
    return out


def linear_constraint_poly(u_coeffs: Sequence[Sequence[int]], r_a_rows: Sequence[Sequence[int]], k: int) -> List[int]:
    """src/ligero/mod.rs:723-736: r_polys = small_domain.ifft(row) for each k-chunk of
    r_a = A.row_mul(r_linear); result = sum_i u_polys[i] * r_polys[i].  Returned zero-padded to
    2k coefficients (the reference's DensePolynomial trims trailing zeros)."""
    w = domain_generator(k)
    acc = [0] * (2 * k)
    for u, ra in zip(u_coeffs, r_a_rows):
        rp = intt(list(ra), w)
        for i, v in enumerate(poly_mul(u, rp)):
            acc[i] = (acc[i] + v) % P
    return acc


def quadratic_constraint_poly(u_coeffs: Sequence[Sequence[int]], r: Sequence[int], m: int, k: int) -> List[int]:
    """src/ligero/mod.rs:842-848: sum_i ((p_x_i * p_y_i) - p_z_i) * r_i over the first 3m rows
    split as [X; Y; Z].  Zero-padded to 2k coefficients."""
    acc = [0] * (2 * k)
    for i in range(m):
        prod = poly_mul(u_coeffs[i], u_coeffs[m + i])
        for j in range(len(prod)):
            z = u_coeffs[2 * m + i][j] if j < k else 0
            acc[j] = (acc[j] + (prod[j] - z) * r[i]) % P
    return acc
