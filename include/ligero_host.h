/*
 * ligero_host.h -- C ABI of the host-side input pipeline (no GPU): the C++ mirror
 * (ligero_amd/host/circuit.hpp) of what sits in front of the encode-and-commit path in
 * NP-Eng/ligero, so that the reference's own fixtures (.r1cs + witness) can be taken to the
 * matrix `preenc_u` and to the linear-test operand r_a without the reference's toolchain.
 *
 *   circuit builders            src/arithmetic_circuit/mod.rs:65-239
 *   lgh_circuit_from_r1cs       src/reader.rs:6-19 + ArithmeticCircuit::from_constraint_system (mod.rs:455-520)
 *   lgh_instance_new            LigeroCircuit::new (src/ligero/mod.rs:147-228, 275-433): insert_one / bump_index,
 *                               dimensions m, k, n, t, the sparse constraint matrix A
 *   lgh_build_preenc            prove + prove_inner up to preenc_u (src/ligero/mod.rs:449-452, 476-516)
 *   lgh_a_row_mul               SparseMatrix::row_mul (src/matrices/mod.rs:100-110) as called at mod.rs:722
 *
 * Field elements: BN254 Fr, 4 x u64 LE limbs, Montgomery form (as in ligero_hip.h).
 * Where the reference panics these functions return LGH_ERR_PANIC and lgh_last_error() carries
 * the message.
 */
#ifndef LIGERO_HOST_H
#define LIGERO_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lgh_circuit lgh_circuit;
typedef struct lgh_instance lgh_instance;

enum { LGH_OK = 0, LGH_ERR_BAD_ARG = -1, LGH_ERR_PANIC = -2, LGH_ERR_OOM = -3 };

const char* lgh_last_error(void);

/* ArithmeticCircuit by hand; node-returning calls give the new node's index (>= 0) or an error (< 0) */
lgh_circuit* lgh_circuit_new(void);
void lgh_circuit_destroy(lgh_circuit* c);
int64_t lgh_circuit_num_nodes(const lgh_circuit* c);
int64_t lgh_constant(lgh_circuit* c, const uint64_t value[4]);
int64_t lgh_new_variable(lgh_circuit* c);
/* new_variable_with_label / get_variable (src/arithmetic_circuit/mod.rs:92-100, 115-117); a label in use, or one the
 * circuit does not have, is the reference's panic (LGH_ERR_PANIC) */
int64_t lgh_new_variable_with_label(lgh_circuit* c, const char* label);
int64_t lgh_get_variable(const lgh_circuit* c, const char* label);
int64_t lgh_circuit_num_gates(const lgh_circuit* c);
int64_t lgh_add(lgh_circuit* c, uint64_t left, uint64_t right);
int64_t lgh_mul(lgh_circuit* c, uint64_t left, uint64_t right);
int64_t lgh_pow(lgh_circuit* c, uint64_t node, uint64_t exponent);
int64_t lgh_minus(lgh_circuit* c, uint64_t node);
/* pow_bigint (mod.rs:164-179; exponent = nlimbs little-endian u64), indicator x^(p-1) (mod.rs:203-217),
 * scalar_product (mod.rs:228-239), mul_nodes (mod.rs:156-161) */
int64_t lgh_pow_bigint(lgh_circuit* c, uint64_t node, const uint64_t* exponent_limbs, uint64_t nlimbs);
int64_t lgh_indicator(lgh_circuit* c, uint64_t node);
int64_t lgh_scalar_product(lgh_circuit* c, const uint64_t* left, const uint64_t* right, uint64_t count);
int64_t lgh_mul_nodes(lgh_circuit* c, const uint64_t* nodes, uint64_t count);
/* evaluate_multioutput (mod.rs:381-387): values of the output nodes in node order (each node once); values_out has
 * room for n_outputs elements, *count_out says how many were written.  Only what the outputs depend on is evaluated. */
int lgh_evaluate_multioutput(const lgh_circuit* c, const uint64_t* node_idx, const uint64_t* values, uint64_t count,
                             const uint64_t* outputs, uint64_t n_outputs, uint64_t* values_out, uint64_t* count_out);

/* one node of the circuit: kind 0 Variable / 1 Constant / 2 Add / 3 Mul (src/arithmetic_circuit/mod.rs:14-24); left/right
 * for gates, value (Montgomery) for constants, the label (NUL-terminated, cut to label_capacity) for variables; any out
 * pointer may be NULL */
int lgh_circuit_node(const lgh_circuit* c, uint64_t index, uint32_t* kind, uint64_t* left, uint64_t* right, uint64_t value[4],
                     char* label, uint64_t label_capacity);

/* Expression front end (src/expression/mod.rs; ligero_amd/host/expression.hpp): handles are shared sub-expressions --
 * using one handle twice is using the same node twice.  Constructors return NULL on error (lgh_last_error). */
typedef struct lgh_expr lgh_expr;
lgh_expr* lgh_expr_variable(const char* label);
lgh_expr* lgh_expr_constant(const uint64_t value[4]);
lgh_expr* lgh_expr_add(const lgh_expr* a, const lgh_expr* b);
lgh_expr* lgh_expr_mul(const lgh_expr* a, const lgh_expr* b);
lgh_expr* lgh_expr_sub(const lgh_expr* a, const lgh_expr* b);   /* a + (-1) * b, mod.rs:209-215 */
lgh_expr* lgh_expr_neg(const lgh_expr* a);
lgh_expr* lgh_expr_pow(const lgh_expr* a, uint64_t exponent);
void lgh_expr_destroy(lgh_expr* e);
/* to_arithmetic_circuit (mod.rs:59-107): the root is the circuit's last node */
int lgh_expr_to_circuit(const lgh_expr* e, lgh_circuit** out);

/* read a circom .r1cs (v1, BN254) and compile it; the output nodes are kept with the circuit */
int lgh_circuit_from_r1cs(lgh_circuit** out, const char* r1cs_path);
int64_t lgh_circuit_num_outputs(const lgh_circuit* c);
int lgh_circuit_outputs(const lgh_circuit* c, uint64_t* outputs_out);

/* LigeroCircuit::new(circuit, outputs, lambda); the circuit is copied */
int lgh_instance_new(lgh_instance** out, const lgh_circuit* c, const uint64_t* outputs, uint64_t n_outputs, uint32_t lambda);
void lgh_instance_destroy(lgh_instance* inst);
/* info_out = { m, k, n, t, num_nodes, num_constants, num_outputs, nnz(A) } */
int lgh_instance_info(const lgh_instance* inst, uint64_t info_out[8]);

/* var assignment (ORIGINAL node indices, as given to LigeroCircuit::prove) -> preenc_u, 4m * k
 * elements row-major; *all_outputs_one (may be NULL) tells whether every output evaluated to 1 */
int lgh_build_preenc(const lgh_instance* inst, const uint64_t* node_idx, const uint64_t* values, uint64_t count,
                     uint64_t* preenc_out, int* all_outputs_one);

/* the same for prove_with_labels (src/ligero/mod.rs:580-611): `labels` = count NUL-terminated strings; an unknown
 * label is "Variable not found: <label>" (LGH_ERR_PANIC) */
int lgh_build_preenc_with_labels(const lgh_instance* inst, const char* const* labels, const uint64_t* values, uint64_t count,
                                 uint64_t* preenc_out, int* all_outputs_one);

/* a1 on the device (include/ligero_hip.h lg_upload_gate_map / lg_encode_commit_from_witness): preenc_u = w + wiring.
 *   lgh_gate_map   sizes first (left / right / constants NULL): *npos_out = positions of the solution vector, *nconst_out =
 *                  constants without a position; then the map itself: left[p] / right[p] = 0xffffffff (no Mul gate at p),
 *                  0x80000000 | c (operand = constants[c]) or the operand's position
 *   lgh_build_w    the W block alone: m * k elements (w zero padded), same assignment convention as lgh_build_preenc */
int lgh_gate_map(const lgh_instance* inst, uint64_t* npos_out, uint64_t* nconst_out, uint32_t* left, uint32_t* right, uint64_t* constants);
int lgh_build_w(const lgh_instance* inst, const uint64_t* node_idx, const uint64_t* values, uint64_t count, uint64_t* w_out, int* all_outputs_one);

/* f3 on the device (include/ligero_hip.h lg_upload_trace_program / lg_encode_commit_from_inputs): the evaluation trace
 * (src/arithmetic_circuit/mod.rs:325-358) as a program over the positions of w, scheduled by dependency level.
 *   lgh_trace_program   sizes first (every array NULL): sizes[0] = npos, [1] = constants, [2] = gates (= length of order),
 *                       [3] = levels, [4] = outputs, [5] = variables (inputs), [6] = formatted nodes (= length of pos_of_node);
 *                       then the arrays: op[npos] (0 input, 1 add, 2 mul, 3 the leading one), left / right [npos] (gates: a position or
 *                       0x80000000 | constant index; otherwise 0xffffffff), constants (4 words each, Montgomery form), order[gates]
 *                       (positions level by level), level_off[levels + 1], outputs[], pos_of_node[nodes] (0xffffffff: no position)
 *   lgh_input_positions the assignment convention of lgh_build_preenc (ORIGINAL node indices) -> positions of w, for
 *                       lg_encode_commit_from_inputs; LGH_ERR_PANIC with the reference's wording for a non-variable node */
int lgh_trace_program(const lgh_instance* inst, uint64_t sizes[7], uint8_t* op, uint32_t* left, uint32_t* right, uint64_t* constants, uint32_t* order,
                      uint64_t* level_off, uint32_t* outputs, uint32_t* pos_of_node);
int lgh_input_positions(const lgh_instance* inst, const uint64_t* node_idx, uint64_t count, uint32_t* positions_out);

/* r_a = A.row_mul(r): r and out have 4 * m * k elements */
int lgh_a_row_mul(const lgh_instance* inst, const uint64_t* r, uint64_t* out);
/* COO dump of A (nnz entries each), row-major order */
int lgh_a_entries(const lgh_instance* inst, uint64_t* row_idx, uint64_t* col_idx, uint64_t* values);

/* Witness file -> wire values (wire 0 first), Montgomery form: circom's witness.json (array of decimal strings, what
 * src/ligero/tests.rs:384-390 reads) or snarkjs' binary .wtns.  *count_out = number of values in the file; at most
 * `capacity` of them are stored (call with capacity 0 to size the buffer). */
int lgh_read_witness(const char* path, uint64_t* values_out, uint64_t capacity, uint64_t* count_out);

/*
 * Fiat-Shamir pieces (ligero_amd/host/transcript.hpp): restated from the published algorithms of
 * un-vendored crates -- PARITY UNPINNED except the ChaCha block function (RFC 8439 vector).
 *   lgh_chacha_block                 the block function with `rounds` in {8, 12, 20}
 *   lgh_field_elements_from_seed     get_field_elements_from_prng   (src/utils.rs:23-29)
 *   lgh_distinct_indices_from_seed   get_distinct_indices_from_prng (src/utils.rs:31-55); *count_out = t
 *   lgh_sponge_*                     PoseidonSponge of test_sponge(): absorb(&Vec<u8>), absorb(&Vec<F>),
 *                                    squeeze_bytes, squeeze_native_field_elements
 */
typedef struct lgh_sponge lgh_sponge;
void lgh_chacha_block(uint32_t rounds, const uint32_t key[8], const uint32_t words12_15[4], uint32_t out[16]);
int lgh_field_elements_from_seed(const uint8_t seed[32], uint64_t n, uint64_t* out);
int lgh_distinct_indices_from_seed(const uint8_t seed[32], uint64_t n, uint64_t t, uint64_t* out, uint64_t* count_out);
lgh_sponge* lgh_sponge_new(void);
void lgh_sponge_destroy(lgh_sponge* s);
int lgh_sponge_absorb_bytes(lgh_sponge* s, const uint8_t* data, uint64_t len);
int lgh_sponge_absorb_elements(lgh_sponge* s, const uint64_t* elems, uint64_t count);
int lgh_sponge_squeeze_bytes(lgh_sponge* s, uint64_t n, uint8_t* out);
int lgh_sponge_squeeze_elements(lgh_sponge* s, uint64_t n, uint64_t* out);
/* absorb_elements on EIGHT sponges at once: elems = 8 * count elements, sponge-major.  Where the host has AVX-512 IFMA and the
 * sponges are in step the eight states advance on the lanes of one vector (ligero_amd/host/poseidon_ifma.hpp); the states are
 * the same as after eight lgh_sponge_absorb_elements calls either way.  lgh_ifma_available: 1 if the vector path can run here. */
int lgh_sponge_absorb_elements_x8(lgh_sponge* const sponges[8], const uint64_t* elems, uint64_t count);
int lgh_ifma_available(void);

#ifdef __cplusplus
}
#endif
#endif /* LIGERO_HOST_H */
