"""ctypes mirror of include/ligero_host.h (ligero_amd/lib/libligero_host.so): the C++ host
side in front of the encode-and-commit path -- circuit builders, circom .r1cs ->
ArithmeticCircuit (src/arithmetic_circuit/mod.rs:455-520), LigeroCircuit::new
(src/ligero/mod.rs:147-433: dimensions, constraint matrix A), witness -> preenc_u
(mod.rs:476-516) and A.row_mul (src/matrices/mod.rs:100-110).  No GPU involved.

Field elements are numpy uint64 (..., 4) arrays in Montgomery form, as everywhere else.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libligero_host.so")

SYMBOLS = [
    "lgh_last_error", "lgh_circuit_new", "lgh_circuit_destroy", "lgh_circuit_num_nodes", "lgh_constant", "lgh_new_variable",
    "lgh_add", "lgh_mul", "lgh_pow", "lgh_minus", "lgh_circuit_from_r1cs", "lgh_circuit_num_outputs", "lgh_circuit_outputs",
    "lgh_instance_new", "lgh_instance_destroy", "lgh_instance_info", "lgh_build_preenc", "lgh_gate_map", "lgh_trace_program", "lgh_input_positions", "lgh_build_w", "lgh_a_row_mul", "lgh_a_entries",
    "lgh_read_witness", "lgh_chacha_block", "lgh_field_elements_from_seed", "lgh_distinct_indices_from_seed", "lgh_sponge_new", "lgh_sponge_destroy",
    "lgh_sponge_absorb_bytes", "lgh_sponge_absorb_elements", "lgh_sponge_squeeze_bytes", "lgh_sponge_squeeze_elements",
    "lgh_new_variable_with_label", "lgh_get_variable", "lgh_circuit_num_gates", "lgh_pow_bigint", "lgh_indicator", "lgh_scalar_product",
    "lgh_mul_nodes", "lgh_evaluate_multioutput", "lgh_build_preenc_with_labels", "lgh_circuit_node",
    "lgh_expr_variable", "lgh_expr_constant", "lgh_expr_add", "lgh_expr_mul", "lgh_expr_sub", "lgh_expr_neg", "lgh_expr_pow",
    "lgh_expr_destroy", "lgh_expr_to_circuit", "lgh_sponge_absorb_elements_x8", "lgh_ifma_available",
]

_vp, _u64, _i64, _u32, _int = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int
_lib = None


class HostPanic(RuntimeError):
    """the reference would have panicked here (message = the reference's)"""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C ligero_amd/host`")
        L = ctypes.CDLL(LIB_PATH)
        L.lgh_last_error.restype = ctypes.c_char_p
        L.lgh_circuit_new.restype = _vp
        L.lgh_circuit_destroy.argtypes = [_vp]
        L.lgh_circuit_destroy.restype = None
        for name, args in (("lgh_circuit_num_nodes", [_vp]), ("lgh_constant", [_vp, _vp]), ("lgh_new_variable", [_vp]),
                           ("lgh_add", [_vp, _u64, _u64]), ("lgh_mul", [_vp, _u64, _u64]), ("lgh_pow", [_vp, _u64, _u64]),
                           ("lgh_minus", [_vp, _u64]), ("lgh_circuit_num_outputs", [_vp]),
                           ("lgh_new_variable_with_label", [_vp, ctypes.c_char_p]), ("lgh_get_variable", [_vp, ctypes.c_char_p]),
                           ("lgh_circuit_num_gates", [_vp]), ("lgh_pow_bigint", [_vp, _u64, _vp, _u64]), ("lgh_indicator", [_vp, _u64]),
                           ("lgh_scalar_product", [_vp, _vp, _vp, _u64]), ("lgh_mul_nodes", [_vp, _vp, _u64])):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = _i64
        L.lgh_circuit_from_r1cs.argtypes = [ctypes.POINTER(_vp), ctypes.c_char_p]
        L.lgh_circuit_node.argtypes = [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _u64]
        for name, args in (("lgh_expr_variable", [ctypes.c_char_p]), ("lgh_expr_constant", [_vp]), ("lgh_expr_add", [_vp, _vp]),
                           ("lgh_expr_mul", [_vp, _vp]), ("lgh_expr_sub", [_vp, _vp]), ("lgh_expr_neg", [_vp]), ("lgh_expr_pow", [_vp, _u64])):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = _vp
        L.lgh_expr_destroy.argtypes = [_vp]
        L.lgh_expr_destroy.restype = None
        L.lgh_expr_to_circuit.argtypes = [_vp, ctypes.POINTER(_vp)]
        L.lgh_circuit_outputs.argtypes = [_vp, _vp]
        L.lgh_instance_new.argtypes = [ctypes.POINTER(_vp), _vp, _vp, _u64, _u32]
        L.lgh_instance_destroy.argtypes = [_vp]
        L.lgh_instance_destroy.restype = None
        L.lgh_instance_info.argtypes = [_vp, _vp]
        L.lgh_build_preenc.argtypes = [_vp, _vp, _vp, _u64, _vp, _vp]
        L.lgh_gate_map.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
        L.lgh_trace_program.argtypes = [_vp] * 10
        L.lgh_input_positions.argtypes = [_vp, _vp, ctypes.c_uint64, _vp]
        L.lgh_build_w.argtypes = [_vp, _vp, _vp, _u64, _vp, _vp]
        L.lgh_build_preenc_with_labels.argtypes = [_vp, _vp, _vp, _u64, _vp, _vp]
        L.lgh_evaluate_multioutput.argtypes = [_vp, _vp, _vp, _u64, _vp, _u64, _vp, _vp]
        L.lgh_a_row_mul.argtypes = [_vp, _vp, _vp]
        L.lgh_a_entries.argtypes = [_vp, _vp, _vp, _vp]
        L.lgh_read_witness.argtypes = [ctypes.c_char_p, _vp, _u64, _vp]
        L.lgh_chacha_block.argtypes = [_u32, _vp, _vp, _vp]
        L.lgh_chacha_block.restype = None
        L.lgh_field_elements_from_seed.argtypes = [_vp, _u64, _vp]
        L.lgh_distinct_indices_from_seed.argtypes = [_vp, _u64, _u64, _vp, _vp]
        L.lgh_sponge_new.restype = _vp
        L.lgh_sponge_destroy.argtypes = [_vp]
        L.lgh_sponge_destroy.restype = None
        L.lgh_sponge_absorb_bytes.argtypes = [_vp, _vp, _u64]
        L.lgh_sponge_absorb_elements.argtypes = [_vp, _vp, _u64]
        L.lgh_sponge_squeeze_bytes.argtypes = [_vp, _u64, _vp]
        L.lgh_sponge_absorb_elements_x8.argtypes = [_vp, _vp, _u64]
        L.lgh_sponge_squeeze_elements.argtypes = [_vp, _u64, _vp]
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        msg = lib().lgh_last_error().decode()
        if rc == -2:
            raise HostPanic(f"{what}: {msg}")
        raise RuntimeError(f"{what}: status {rc} ({msg})")
    return rc


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def c_labels(labels: Sequence[str]):
    """list of str -> a `const char* const*` argument (keep the returned array alive for the call)"""
    arr = (ctypes.c_char_p * len(labels))(*[s.encode() for s in labels])
    return arr


class ArithmeticCircuit:
    """src/arithmetic_circuit/mod.rs: builders + from_constraint_system"""

    def __init__(self, _handle=None):
        self._L = lib()
        self._h = _vp(_handle) if _handle is not None else _vp(self._L.lgh_circuit_new())
        self.outputs = []

    @classmethod
    def from_r1cs(cls, path: str) -> "ArithmeticCircuit":
        L = lib()
        h = _vp()
        _check(L.lgh_circuit_from_r1cs(ctypes.byref(h), path.encode()), "from_constraint_system")
        c = cls(h.value)
        n = _check(L.lgh_circuit_num_outputs(c._h), "num_outputs")
        out = np.zeros(n, dtype=np.uint64)
        _check(L.lgh_circuit_outputs(c._h, _p(out)), "outputs")
        c.outputs = [int(x) for x in out]
        return c

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lgh_circuit_destroy(self._h)
            self._h = None

    def num_nodes(self) -> int:
        return _check(self._L.lgh_circuit_num_nodes(self._h), "num_nodes")

    def last(self) -> int:
        return self.num_nodes() - 1

    def node(self, i: int):
        """("Variable", label) / ("Constant", limbs) / ("Add", l, r) / ("Mul", l, r)"""
        kind, l, r = ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        val = np.zeros(4, dtype=np.uint64)
        label = ctypes.create_string_buffer(256)
        _check(self._L.lgh_circuit_node(self._h, i, ctypes.cast(ctypes.byref(kind), _vp), ctypes.cast(ctypes.byref(l), _vp),
                                        ctypes.cast(ctypes.byref(r), _vp), _p(val), ctypes.cast(label, _vp), 256), "node")
        if kind.value == 0:
            return ("Variable", label.value.decode())
        if kind.value == 1:
            return ("Constant", val)
        return ("Add" if kind.value == 2 else "Mul", l.value, r.value)

    def constant(self, value_mont: np.ndarray) -> int:
        v = np.ascontiguousarray(value_mont, dtype=np.uint64).reshape(4)
        return _check(self._L.lgh_constant(self._h, _p(v)), "constant")

    def new_variable(self) -> int:
        return _check(self._L.lgh_new_variable(self._h), "new_variable")

    def new_variable_with_label(self, label: str) -> int:
        return _check(self._L.lgh_new_variable_with_label(self._h, label.encode()), "new_variable_with_label")

    def new_variables(self, num: int):
        return [self.new_variable() for _ in range(num)]

    def get_variable(self, label: str) -> int:
        return _check(self._L.lgh_get_variable(self._h, label.encode()), "get_variable")

    def num_gates(self) -> int:
        return _check(self._L.lgh_circuit_num_gates(self._h), "num_gates")

    def add(self, l: int, r: int) -> int:
        return _check(self._L.lgh_add(self._h, l, r), "add")

    def add_nodes(self, nodes: Sequence[int]) -> int:
        acc = nodes[0]
        for i in nodes[1:]:
            acc = self.add(acc, i)
        return acc

    def mul_nodes(self, nodes: Sequence[int]) -> int:
        a = np.ascontiguousarray(nodes, dtype=np.uint64)
        return _check(self._L.lgh_mul_nodes(self._h, _p(a), a.shape[0]), "mul_nodes")

    def pow_bigint(self, node: int, exponent: int) -> int:
        nl = max(1, (exponent.bit_length() + 63) // 64)
        limbs = np.array([(exponent >> (64 * i)) & (2**64 - 1) for i in range(nl)], dtype=np.uint64)
        return _check(self._L.lgh_pow_bigint(self._h, node, _p(limbs), nl), "pow_bigint")

    def indicator(self, node: int) -> int:
        return _check(self._L.lgh_indicator(self._h, node), "indicator")

    def scalar_product(self, left: Sequence[int], right: Sequence[int]) -> int:
        n = min(len(left), len(right))
        a = np.ascontiguousarray(left[:n], dtype=np.uint64)
        b = np.ascontiguousarray(right[:n], dtype=np.uint64)
        return _check(self._L.lgh_scalar_product(self._h, _p(a), _p(b), n), "scalar_product")

    def evaluate_multioutput(self, node_idx: Sequence[int], values_mont: np.ndarray, outputs: Sequence[int]) -> np.ndarray:
        """evaluate_multioutput (mod.rs:381-387): output values in node order, (count, 4) Montgomery limbs"""
        idx = np.ascontiguousarray(node_idx, dtype=np.uint64)
        vals = np.ascontiguousarray(values_mont, dtype=np.uint64).reshape(idx.shape[0], 4)
        outs = np.ascontiguousarray(outputs, dtype=np.uint64)
        res = np.zeros((outs.shape[0], 4), dtype=np.uint64)
        cnt = ctypes.c_uint64(0)
        _check(self._L.lgh_evaluate_multioutput(self._h, _p(idx), _p(vals), idx.shape[0], _p(outs), outs.shape[0], _p(res),
                                                ctypes.cast(ctypes.byref(cnt), _vp)), "evaluate_multioutput")
        return res[:cnt.value]

    def evaluate_node(self, node_idx: Sequence[int], values_mont: np.ndarray, node: int) -> np.ndarray:
        return self.evaluate_multioutput(node_idx, values_mont, [node])[0]

    def mul(self, l: int, r: int) -> int:
        return _check(self._L.lgh_mul(self._h, l, r), "mul")

    def pow(self, node: int, e: int) -> int:
        return _check(self._L.lgh_pow(self._h, node, e), "pow")

    def minus(self, node: int) -> int:
        return _check(self._L.lgh_minus(self._h, node), "minus")


P_BN254_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_R_BN254_FR = (1 << 256) % P_BN254_FR


def fr_mont(value: int) -> np.ndarray:
    """F::from(value) for BN254 Fr as the four Montgomery limbs the libraries exchange"""
    v = (value % P_BN254_FR) * _R_BN254_FR % P_BN254_FR
    return np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


class Expression:
    """src/expression/mod.rs over BN254 Fr: variables and constants combined with + - * and pow into a shared DAG.  As in
    the reference, identity is by node: `x = Expression.variable("x"); x * x` uses ONE variable node twice, while two
    Expression.variable("x") calls are two nodes.  Python ints on either side of an operator are F::from(int) constants."""

    def __init__(self, handle):
        self._L = lib()
        if not handle:
            raise HostPanic("expression: " + self._L.lgh_last_error().decode())
        self._h = _vp(handle)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lgh_expr_destroy(self._h)
            self._h = None

    @classmethod
    def variable(cls, label: str) -> "Expression":
        return cls(lib().lgh_expr_variable(label.encode()))

    @classmethod
    def constant(cls, value) -> "Expression":
        """value: a Python int (F::from) or four Montgomery limbs"""
        v = fr_mont(value) if isinstance(value, int) else np.ascontiguousarray(value, dtype=np.uint64).reshape(4)
        return cls(lib().lgh_expr_constant(_p(v)))

    @staticmethod
    def _lift(x) -> "Expression":
        return x if isinstance(x, Expression) else Expression.constant(x)

    def __add__(self, o):
        rhs = Expression._lift(o)              # kept alive across the call: a temporary's handle dies with it
        return Expression(self._L.lgh_expr_add(self._h, rhs._h))

    def __radd__(self, o):
        lhs = Expression._lift(o)
        return Expression(self._L.lgh_expr_add(lhs._h, self._h))

    def __mul__(self, o):
        rhs = Expression._lift(o)              # kept alive across the call: a temporary's handle dies with it
        return Expression(self._L.lgh_expr_mul(self._h, rhs._h))

    def __rmul__(self, o):
        lhs = Expression._lift(o)
        return Expression(self._L.lgh_expr_mul(lhs._h, self._h))

    def __sub__(self, o):
        rhs = Expression._lift(o)              # kept alive across the call: a temporary's handle dies with it
        return Expression(self._L.lgh_expr_sub(self._h, rhs._h))

    def __rsub__(self, o):
        lhs = Expression._lift(o)
        return Expression(self._L.lgh_expr_sub(lhs._h, self._h))

    def __neg__(self):
        return Expression(self._L.lgh_expr_neg(self._h))

    def pow(self, e: int) -> "Expression":
        return Expression(self._L.lgh_expr_pow(self._h, e))

    __pow__ = pow

    def to_arithmetic_circuit(self) -> "ArithmeticCircuit":
        h = _vp()
        _check(self._L.lgh_expr_to_circuit(self._h, ctypes.byref(h)), "to_arithmetic_circuit")
        return ArithmeticCircuit(h.value)


class LigeroInstance:
    """LigeroCircuit::new (src/ligero/mod.rs:147-228): dimensions + constraint matrix A"""

    def __init__(self, circuit: ArithmeticCircuit, outputs: Optional[Sequence[int]] = None, lam: int = 128):
        self._L = lib()
        outs = np.ascontiguousarray(circuit.outputs if outputs is None else outputs, dtype=np.uint64)
        self._h = _vp()
        _check(self._L.lgh_instance_new(ctypes.byref(self._h), circuit._h, _p(outs), outs.shape[0], lam), "LigeroCircuit::new")
        info = np.zeros(8, dtype=np.uint64)
        _check(self._L.lgh_instance_info(self._h, _p(info)), "info")
        (self.m, self.k, self.n, self.t, self.num_nodes, self.num_constants, self.num_outputs, self.a_nnz) = (int(x) for x in info)
        self.rows = 4 * self.m

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lgh_instance_destroy(self._h)
            self._h = None

    def build_preenc_u(self, node_idx: Sequence[int], values_mont: np.ndarray) -> Tuple[np.ndarray, bool]:
        """prove + prove_inner up to preenc_u (mod.rs:449-452, 476-516).  Returns ((4m, k, 4), all_outputs_one)."""
        idx = np.ascontiguousarray(node_idx, dtype=np.uint64)
        vals = np.ascontiguousarray(values_mont, dtype=np.uint64).reshape(idx.shape[0], 4)
        out = np.empty((self.rows, self.k, 4), dtype=np.uint64)
        ok = _int(0)
        _check(self._L.lgh_build_preenc(self._h, _p(idx), _p(vals), idx.shape[0], _p(out), ctypes.cast(ctypes.byref(ok), _vp)), "prove_inner")
        return out, bool(ok.value)

    def trace_program(self):
        """the evaluation trace as a level-scheduled program over the positions of w (lgh_trace_program), for
        lg_upload_trace_program: dict of op, left, right, constants, order, level_off, outputs, pos_of_node, num_inputs"""
        sizes = np.zeros(7, dtype=np.uint64)
        _check(self._L.lgh_trace_program(self._h, _p(sizes), None, None, None, None, None, None, None, None), "trace_program")
        npos, nconst, ngates, nlev, nout, nin, nnodes = (int(x) for x in sizes)
        t = {"op": np.zeros(npos, dtype=np.uint8), "left": np.zeros(npos, dtype=np.uint32), "right": np.zeros(npos, dtype=np.uint32),
             "constants": np.zeros((nconst, 4), dtype=np.uint64), "order": np.zeros(ngates, dtype=np.uint32),
             "level_off": np.zeros(nlev + 1, dtype=np.uint64), "outputs": np.zeros(nout, dtype=np.uint32),
             "pos_of_node": np.zeros(nnodes, dtype=np.uint32), "num_inputs": nin}
        _check(self._L.lgh_trace_program(self._h, _p(sizes), _p(t["op"]), _p(t["left"]), _p(t["right"]), _p(t["constants"]), _p(t["order"]),
                                         _p(t["level_off"]), _p(t["outputs"]), _p(t["pos_of_node"])), "trace_program")
        return t

    def input_positions(self, node_idx):
        """positions of w for an assignment given by ORIGINAL node indices (the convention of build_preenc_u)"""
        idx = np.ascontiguousarray(node_idx, dtype=np.uint64)
        out = np.zeros(idx.shape[0], dtype=np.uint32)
        _check(self._L.lgh_input_positions(self._h, _p(idx), idx.shape[0], _p(out)), "input_positions")
        return out

    def gate_map(self):
        """the circuit's wiring for lg_upload_gate_map: (left, right) uint32 arrays over the positions of the solution vector
        (0xffffffff: no Mul gate; 0x80000000 | c: constants[c]; else a position of w) and the constants (c, 4)"""
        npos, nconst = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self._L.lgh_gate_map(self._h, ctypes.cast(ctypes.byref(npos), _vp), ctypes.cast(ctypes.byref(nconst), _vp), None, None, None), "gate_map")
        left = np.empty(npos.value, dtype=np.uint32)
        right = np.empty(npos.value, dtype=np.uint32)
        consts = np.empty((nconst.value, 4), dtype=np.uint64)
        _check(self._L.lgh_gate_map(self._h, ctypes.cast(ctypes.byref(npos), _vp), ctypes.cast(ctypes.byref(nconst), _vp), _p(left), _p(right), _p(consts)), "gate_map")
        return left, right, consts

    def build_w(self, node_idx: Sequence[int], values_mont: np.ndarray) -> Tuple[np.ndarray, bool]:
        """the W block of preenc_u alone (mod.rs:483-509: the kept node values, zero padded): ((m, k, 4), all_outputs_one)"""
        idx = np.ascontiguousarray(node_idx, dtype=np.uint64)
        vals = np.ascontiguousarray(values_mont, dtype=np.uint64).reshape(idx.shape[0], 4)
        out = np.empty((self.m, self.k, 4), dtype=np.uint64)
        ok = _int(0)
        _check(self._L.lgh_build_w(self._h, _p(idx), _p(vals), idx.shape[0], _p(out), ctypes.cast(ctypes.byref(ok), _vp)), "prove_inner")
        return out, bool(ok.value)

    def build_preenc_u_with_labels(self, labels: Sequence[str], values_mont: np.ndarray) -> Tuple[np.ndarray, bool]:
        """prove_with_labels + prove_inner up to preenc_u (mod.rs:580-611, 476-516)"""
        vals = np.ascontiguousarray(values_mont, dtype=np.uint64).reshape(len(labels), 4)
        out = np.empty((self.rows, self.k, 4), dtype=np.uint64)
        ok = _int(0)
        arr = c_labels(labels)
        _check(self._L.lgh_build_preenc_with_labels(self._h, ctypes.cast(arr, _vp), _p(vals), len(labels), _p(out),
                                                    ctypes.cast(ctypes.byref(ok), _vp)), "prove_inner")
        return out, bool(ok.value)

    def a_row_mul(self, r_mont: np.ndarray) -> np.ndarray:
        """self.a.row_mul(&r_linear) (mod.rs:722): (4mk, 4) -> (4mk, 4)"""
        r = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(self.rows * self.k, 4)
        out = np.empty_like(r)
        _check(self._L.lgh_a_row_mul(self._h, _p(r), _p(out)), "A.row_mul")
        return out

    def a_entries(self):
        """COO dump of A: (row, col, value[4]) arrays"""
        rows = np.empty(self.a_nnz, dtype=np.uint64)
        cols = np.empty(self.a_nnz, dtype=np.uint64)
        vals = np.empty((self.a_nnz, 4), dtype=np.uint64)
        _check(self._L.lgh_a_entries(self._h, _p(rows), _p(cols), _p(vals)), "a_entries")
        return rows, cols, vals


def read_witness(path: str) -> np.ndarray:
    """witness.json (decimal strings) or .wtns -> (wires, 4) Montgomery limbs, wire 0 first"""
    cnt = ctypes.c_uint64(0)
    _check(lib().lgh_read_witness(path.encode(), None, 0, ctypes.cast(ctypes.byref(cnt), _vp)), "read_witness")
    out = np.empty((cnt.value, 4), dtype=np.uint64)
    _check(lib().lgh_read_witness(path.encode(), _p(out), cnt.value, ctypes.cast(ctypes.byref(cnt), _vp)), "read_witness")
    return out


# ---- Fiat-Shamir pieces (ligero_amd/host/transcript.hpp; PARITY UNPINNED, see there)
def chacha_block(rounds: int, key_words, words12_15) -> np.ndarray:
    key = np.ascontiguousarray(key_words, dtype=np.uint32)
    w = np.ascontiguousarray(words12_15, dtype=np.uint32)
    out = np.empty(16, dtype=np.uint32)
    lib().lgh_chacha_block(rounds, _p(key), _p(w), _p(out))
    return out


def field_elements_from_seed(seed: bytes, n: int) -> np.ndarray:
    """get_field_elements_from_prng (src/utils.rs:23-29): (n, 4) Montgomery limbs"""
    s = np.frombuffer(seed, dtype=np.uint8).copy()
    out = np.empty((n, 4), dtype=np.uint64)
    _check(lib().lgh_field_elements_from_seed(_p(s), n, _p(out)), "get_field_elements_from_prng")
    return out


def distinct_indices_from_seed(seed: bytes, n: int, t: int) -> np.ndarray:
    """get_distinct_indices_from_prng (src/utils.rs:31-55)"""
    s = np.frombuffer(seed, dtype=np.uint8).copy()
    out = np.empty(max(t, 1), dtype=np.uint64)
    cnt = ctypes.c_uint64(0)
    _check(lib().lgh_distinct_indices_from_seed(_p(s), n, t, _p(out), ctypes.cast(ctypes.byref(cnt), _vp)), "get_distinct_indices_from_prng")
    return out[:cnt.value]


class PoseidonSponge:
    """test_sponge() of ark-poly-commit (src/ligero/tests.rs:151)"""

    def __init__(self):
        self._L = lib()
        self._h = self._L.lgh_sponge_new()
        if not self._h:
            raise MemoryError("lgh_sponge_new")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lgh_sponge_destroy(self._h)
            self._h = None

    def absorb_bytes(self, data: bytes):
        b = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        _check(self._L.lgh_sponge_absorb_bytes(self._h, _p(b) if b.size else None, b.size), "absorb")

    def absorb_elements(self, elems_mont: np.ndarray):
        e = np.ascontiguousarray(elems_mont, dtype=np.uint64).reshape(-1, 4)
        _check(self._L.lgh_sponge_absorb_elements(self._h, _p(e) if e.size else None, e.shape[0]), "absorb")

    def squeeze_bytes(self, n: int) -> bytes:
        out = np.empty(max(n, 1), dtype=np.uint8)
        _check(self._L.lgh_sponge_squeeze_bytes(self._h, n, _p(out)), "squeeze_bytes")
        return out[:n].tobytes()

    def squeeze_elements(self, n: int) -> np.ndarray:
        out = np.empty((max(n, 1), 4), dtype=np.uint64)
        _check(self._L.lgh_sponge_squeeze_elements(self._h, n, _p(out)), "squeeze_native_field_elements")
        return out[:n]


def ifma_available() -> bool:
    """True where the eight-sponges-per-vector path of the transcript (AVX-512 IFMA) can run"""
    return bool(lib().lgh_ifma_available())


def sponges_absorb_elements_x8(sponges: Sequence["PoseidonSponge"], elems_mont: np.ndarray):
    """absorb_elements on eight sponges at once: elems_mont (8, count, 4)"""
    assert len(sponges) == 8
    e = np.ascontiguousarray(elems_mont, dtype=np.uint64).reshape(8, -1, 4)
    handles = (_vp * 8)(*[s._h for s in sponges])
    _check(lib().lgh_sponge_absorb_elements_x8(ctypes.cast(handles, _vp), _p(e) if e.size else None, e.shape[1]), "absorb x8")
