/*
 * oracle/okzg.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the c-kzg-4844 v2.1.8 host algorithms for the MSM/FFT hot path
 * and its callers.  Every function cites the reference file:line it follows.  The arithmetic
 * underneath (oracle/bls12_381.c) restates what the reference takes from blst v0.3.16, which is
 * absent from /root/reference.  Pinned against all 368 consensus-spec vectors of
 * /root/reference/tests (tests/golden/, tests/test_oracle_vectors.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code; the
 * product library (c-kzg-4844_amd/) never links, includes or calls it.
 */
#ifndef ORACLE_OKZG_H
#define ORACLE_OKZG_H

#include "bls12_381.h"
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKZG_FE_PER_BLOB 4096
#define OKZG_FE_PER_EXT_BLOB 8192
#define OKZG_FE_PER_CELL 64
#define OKZG_CELLS_PER_BLOB 64
#define OKZG_CELLS_PER_EXT_BLOB 128
#define OKZG_BYTES_PER_BLOB 131072
#define OKZG_BYTES_PER_CELL 2048

enum { OKZG_OK = 0, OKZG_BADARGS = 1, OKZG_ERROR = 2, OKZG_MALLOC = 3 };

/* field order follows src/setup/settings.h:27-79 */
typedef struct {
    ofr_t *roots_of_unity;
    ofr_t *brp_roots_of_unity;
    ofr_t *reverse_roots_of_unity;
    og1_t *g1_values_monomial;
    og1_t *g1_values_lagrange_brp;
    og2_t *g2_values_monomial;
    og1_t **x_ext_fft_columns;
    og1_affine_t **tables;
    size_t wbits;
    size_t scratch_size;
} OKZGSettings;

/* src/setup/setup.h:31-44 */
int okzg_load_trusted_setup(OKZGSettings *out, const uint8_t *g1_monomial, uint64_t n_g1_mono,
                            const uint8_t *g1_lagrange, uint64_t n_g1_lagr,
                            const uint8_t *g2_monomial, uint64_t n_g2, uint64_t precompute);
int okzg_load_trusted_setup_file(OKZGSettings *out, FILE *in, uint64_t precompute);
void okzg_free_trusted_setup(OKZGSettings *s);

/* src/eip4844/eip4844.h:43-84 */
int okzg_blob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const OKZGSettings *s);
int okzg_compute_kzg_proof(uint8_t proof_out[48], uint8_t y_out[32], const uint8_t *blob,
                           const uint8_t z[32], const OKZGSettings *s);
int okzg_compute_blob_kzg_proof(uint8_t out[48], const uint8_t *blob, const uint8_t commitment[48],
                                const OKZGSettings *s);
int okzg_verify_kzg_proof(bool *ok, const uint8_t commitment[48], const uint8_t z[32],
                          const uint8_t y[32], const uint8_t proof[48], const OKZGSettings *s);
int okzg_verify_blob_kzg_proof(bool *ok, const uint8_t *blob, const uint8_t commitment[48],
                               const uint8_t proof[48], const OKZGSettings *s);
int okzg_verify_blob_kzg_proof_batch(bool *ok, const uint8_t *blobs, const uint8_t *commitments,
                                     const uint8_t *proofs, uint64_t n, const OKZGSettings *s);
void okzg_compute_challenge(ofr_t *out, const uint8_t *blob, const og1_t *commitment);

/* src/eip7594/eip7594.h:35-68 */
int okzg_compute_cells_and_kzg_proofs(uint8_t *cells, uint8_t *proofs, const uint8_t *blob,
                                      const OKZGSettings *s);
int okzg_recover_cells_and_kzg_proofs(uint8_t *recovered_cells, uint8_t *recovered_proofs,
                                      const uint64_t *cell_indices, const uint8_t *cells,
                                      uint64_t num_cells, const OKZGSettings *s);
int okzg_verify_cell_kzg_proof_batch(bool *ok, const uint8_t *commitments,
                                     const uint64_t *cell_indices, const uint8_t *cells,
                                     const uint8_t *proofs, uint64_t num_cells,
                                     const OKZGSettings *s);
int okzg_compute_verify_cell_kzg_proof_batch_challenge(
    ofr_t *out, const uint8_t *commitments, uint64_t num_commitments,
    const uint64_t *commitment_indices, const uint64_t *cell_indices, const uint8_t *cells,
    const uint8_t *proofs, uint64_t num_cells);

/* hot-path building blocks, exposed for kernel-level parity tests */
int okzg_g1_lincomb_fast(og1_t *out, const og1_t *p, const ofr_t *coeffs, size_t len);
void okzg_g1_lincomb_naive(og1_t *out, const og1_t *p, const ofr_t *coeffs, size_t len);
int okzg_fr_fft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s);
int okzg_fr_ifft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s);
int okzg_coset_fft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s);
int okzg_coset_ifft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s);
int okzg_g1_fft(og1_t *out, const og1_t *in, size_t n, const OKZGSettings *s);
int okzg_g1_ifft_unscaled(og1_t *out, const og1_t *in, size_t n, const OKZGSettings *s);
int okzg_bit_reversal_permutation(void *values, size_t size, size_t n);
int okzg_poly_lagrange_to_monomial(ofr_t *out, const ofr_t *lagrange, size_t len,
                                   const OKZGSettings *s);
int okzg_compute_fk20_cell_proofs(og1_t *out, const ofr_t *poly, const OKZGSettings *s);
int okzg_recover_cells(ofr_t *out, const uint64_t *cell_indices, size_t num_cells, ofr_t *cells,
                       const OKZGSettings *s);
int okzg_blob_to_polynomial(ofr_t *p, const uint8_t *blob);
int okzg_evaluate_polynomial_in_evaluation_form(ofr_t *out, const ofr_t *poly, const ofr_t *x,
                                                const OKZGSettings *s);

/* convenience for ctypes: allocate/free a settings struct, sizes */
OKZGSettings *okzg_settings_new(void);
void okzg_settings_delete(OKZGSettings *s);

#ifdef __cplusplus
}
#endif
#endif
