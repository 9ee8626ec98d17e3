// Host side of the generic-field path (generic_kernels.h): interface used by the translation units behind the C ABI (lg_context.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct gf_state;   // device buffers + tables of one context over a generic field

// field: lg_field of include/ligero_hip.h (1 = BLS12-377 Fq, 2 = BN254 Fr through the generic kernels).  Returns an lg_status.
int gf_create(gf_state** out, int field, uint32_t rows, uint32_t k, uint32_t n, uint32_t batch, hipStream_t stream, char* err, size_t errlen);
void gf_destroy(gf_state* g);
uint32_t gf_element_words64(const gf_state* g);
int gf_upload(gf_state* g, const uint64_t* preenc);
int gf_commit(gf_state* g, const uint64_t* host_pre, uint64_t* host_coeffs);   // host_pre == nullptr: resident rows
int gf_sync(gf_state* g);
int gf_read_root(gf_state* g, uint8_t* out);
int gf_read_coeffs(gf_state* g, uint64_t* out);
int gf_read_leaves(gf_state* g, uint8_t* out);
int gf_read_nodes(gf_state* g, uint8_t* out);
int gf_read_codeword_rows(gf_state* g, uint32_t proof, uint32_t row0, uint32_t nrows, uint64_t* out);
int gf_open_columns(gf_state* g, uint32_t proof0, uint32_t nproofs, const uint32_t* idx, uint32_t t, uint64_t* cols, uint8_t* sib, uint8_t* paths);
int gf_reed_solomon(gf_state* g, const uint64_t* in, uint32_t nrows, uint64_t* out, bool interpolate, bool evaluate);
bool gf_committed(const gf_state* g);
// the arithmetic of the three sub-proofs on the resident commitment (batch 1), as the BN254 entry points of the same names
int gf_interleaved_row_mul(gf_state* g, const uint64_t* r, uint64_t* out);
int gf_linear_constraint_poly(gf_state* g, const uint64_t* r_a, uint64_t* coeffs_out);
int gf_quadratic_constraint_poly(gf_state* g, const uint64_t* r, uint64_t* coeffs_out);
