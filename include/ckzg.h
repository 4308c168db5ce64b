/*
 * ckzg.h -- the drop-in C-ABI boundary of the MI355X build.
 *
 * Same symbols, argument meaning, return codes and struct layouts as the public headers of
 * ethereum/c-kzg-4844 v2.1.8 (src/ckzg.h:19-21 -> src/eip4844/eip4844.h, src/eip7594/eip7594.h,
 * src/setup/setup.h and the common/ headers they pull in), so a binding that compiled the
 * reference's ckzg.c + libblst can link libckzg_hip.so instead.  The implementation behind it is
 * new: host logic in C++ and the G1-MSM / Fr-NTT hot path as HIP kernels for gfx950.
 *
 * Every declaration cites the reference declaration it replaces (paths under /root/reference).
 * Additive batch / device-pointer entry points live in ckzg_hip.h.
 */
#ifndef CKZG_H
#define CKZG_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sizes: src/eip4844/blob.h:29-42, src/eip7594/cell.h:28-37, src/common/bytes.h:30-42 ---- */
#define BYTES_PER_COMMITMENT 48
#define BYTES_PER_PROOF 48
#define BYTES_PER_FIELD_ELEMENT 32
#define BITS_PER_FIELD_ELEMENT 255
#define FIELD_ELEMENTS_PER_BLOB 4096
#define BYTES_PER_BLOB (FIELD_ELEMENTS_PER_BLOB * BYTES_PER_FIELD_ELEMENT)
#define FIELD_ELEMENTS_PER_EXT_BLOB (FIELD_ELEMENTS_PER_BLOB * 2)
#define FIELD_ELEMENTS_PER_CELL 64
#define BYTES_PER_CELL (FIELD_ELEMENTS_PER_CELL * BYTES_PER_FIELD_ELEMENT)
#define CELLS_PER_BLOB (FIELD_ELEMENTS_PER_BLOB / FIELD_ELEMENTS_PER_CELL)
#define CELLS_PER_EXT_BLOB (FIELD_ELEMENTS_PER_EXT_BLOB / FIELD_ELEMENTS_PER_CELL)

/* ---- return codes: src/common/ret.h:24-29 ---- */
typedef enum {
    C_KZG_OK = 0,  /* success */
    C_KZG_BADARGS, /* invalid (untrusted) input */
    C_KZG_ERROR,   /* internal error, including: GPU unavailable / device failure */
    C_KZG_MALLOC,  /* host or device allocation failed */
} C_KZG_RET;

/* ---- wire types: src/common/bytes.h:49-56, src/eip4844/blob.h:49-51, src/eip7594/cell.h:44-46,
 *      src/eip4844/eip4844.h:30-33 ---- */
typedef struct { uint8_t bytes[32]; } Bytes32;
typedef struct { uint8_t bytes[48]; } Bytes48;
typedef struct { uint8_t bytes[BYTES_PER_BLOB]; } Blob;
typedef struct { uint8_t bytes[BYTES_PER_CELL]; } Cell;
typedef Bytes48 KZGCommitment;
typedef Bytes48 KZGProof;

/* ---- internal element types; same size, alignment and value encoding (Montgomery form,
 *      little-endian 64-bit limbs) as blst_fr / blst_fp / blst_p1 / blst_p1_affine / blst_p2
 *      behind fr_t, g1_t, g2_t in src/common/fr.h:27 and src/common/ec.h:26-29 ---- */
typedef struct { uint64_t l[4]; } fr_t;
typedef struct { uint64_t l[6]; } ckzg_fp_t;
typedef struct { ckzg_fp_t fp[2]; } ckzg_fp2_t;
typedef struct { ckzg_fp_t x, y, z; } g1_t;
typedef struct { ckzg_fp_t x, y; } ckzg_g1_affine_t;
typedef struct { ckzg_fp2_t x, y, z; } g2_t;

/* ---- KZGSettings: src/setup/settings.h:27-79.  80 bytes on LP64; bindings allocate or embed it
 *      themselves, so the size and field order are ABI.  Field contents are private to the
 *      library.  The GPU context is found through a registry keyed by the roots_of_unity
 *      pointer (device_ctx.hip), so copies/moves of the struct (Go embeds it by value, Rust moves
 *      it) keep working, no field is added, and a struct this library did not load is simply
 *      not found (C_KZG_ERROR). ---- */
typedef struct {
    fr_t *roots_of_unity;               /* w^i, i = 0..8192 */
    fr_t *brp_roots_of_unity;           /* bit-reversed, 8192 */
    fr_t *reverse_roots_of_unity;       /* w^-i, i = 0..8192 */
    g1_t *g1_values_monomial;           /* 4096 */
    g1_t *g1_values_lagrange_brp;       /* 4096, bit-reversed order */
    g2_t *g2_values_monomial;           /* 65 */
    g1_t **x_ext_fft_columns;           /* 128 x 64, FK20 */
    ckzg_g1_affine_t **tables;          /* unused by this build (fixed-base tables live in HBM) */
    size_t wbits;                       /* the `precompute` argument */
    size_t scratch_size;                /* unused by this build */
} KZGSettings;

/* ---- trusted setup: src/setup/setup.h:31-44 ----
 * One check is STRICTER than the reference's: the reference accepts any setup point that lies on the curve
 * (src/setup/setup.c:447-477: blst_p1_uncompress, no subgroup check -- "the file is trusted"); this library also
 * requires every G1 point of the file to lie in the prime-order subgroup (or be the identity) and answers
 * C_KZG_BADARGS otherwise.  Its fixed-base tables and G1-FFT twiddles use the endomorphism (x, y) -> (beta x, y),
 * which equals [lambda]P only on that subgroup: a sum over other points would silently differ from the reference's.
 * The mainnet setup and every setup that is a real KZG ceremony output pass. */
C_KZG_RET load_trusted_setup(KZGSettings *out, const uint8_t *g1_monomial_bytes,
                             uint64_t num_g1_monomial_bytes, const uint8_t *g1_lagrange_bytes,
                             uint64_t num_g1_lagrange_bytes, const uint8_t *g2_monomial_bytes,
                             uint64_t num_g2_monomial_bytes, uint64_t precompute);
C_KZG_RET load_trusted_setup_file(KZGSettings *out, FILE *in, uint64_t precompute);
void free_trusted_setup(KZGSettings *s);

/* ---- EIP-4844: src/eip4844/eip4844.h:43-84 ---- */
C_KZG_RET blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const KZGSettings *s);
C_KZG_RET compute_kzg_proof(KZGProof *proof_out, Bytes32 *y_out, const Blob *blob,
                            const Bytes32 *z_bytes, const KZGSettings *s);
C_KZG_RET compute_blob_kzg_proof(KZGProof *out, const Blob *blob, const Bytes48 *commitment_bytes,
                                 const KZGSettings *s);
C_KZG_RET verify_kzg_proof(bool *ok, const Bytes48 *commitment_bytes, const Bytes32 *z_bytes,
                           const Bytes32 *y_bytes, const Bytes48 *proof_bytes,
                           const KZGSettings *s);
C_KZG_RET verify_blob_kzg_proof(bool *ok, const Blob *blob, const Bytes48 *commitment_bytes,
                                const Bytes48 *proof_bytes, const KZGSettings *s);
C_KZG_RET verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments_bytes,
                                      const Bytes48 *proofs_bytes, uint64_t n,
                                      const KZGSettings *s);
/* test-exposed internal, src/eip4844/eip4844.h:84 */
void compute_challenge(fr_t *eval_challenge_out, const Blob *blob, const g1_t *commitment);

/* ---- EIP-7594: src/eip7594/eip7594.h:35-68 ---- */
C_KZG_RET compute_cells_and_kzg_proofs(Cell *cells, KZGProof *proofs, const Blob *blob,
                                       const KZGSettings *s);
C_KZG_RET recover_cells_and_kzg_proofs(Cell *recovered_cells, KZGProof *recovered_proofs,
                                       const uint64_t *cell_indices, const Cell *cells,
                                       uint64_t num_cells, const KZGSettings *s);
C_KZG_RET verify_cell_kzg_proof_batch(bool *ok, const Bytes48 *commitments_bytes,
                                      const uint64_t *cell_indices, const Cell *cells,
                                      const Bytes48 *proofs_bytes, uint64_t num_cells,
                                      const KZGSettings *s);
/* test-exposed internal, src/eip7594/eip7594.h:59-68 */
C_KZG_RET compute_verify_cell_kzg_proof_batch_challenge(
    fr_t *challenge_out, const Bytes48 *commitments_bytes, uint64_t num_commitments,
    const uint64_t *commitment_indices, const uint64_t *cell_indices, const Cell *cells,
    const Bytes48 *proofs_bytes, uint64_t num_cells);

/* ---- helpers the Go binding also imports (bindings/go/main.go:586,598-601,647;
 *      src/common/bytes.h) ---- */
C_KZG_RET bytes_to_kzg_commitment(g1_t *out, const Bytes48 *b);
C_KZG_RET bytes_to_kzg_proof(g1_t *out, const Bytes48 *b);
C_KZG_RET bytes_to_bls_field(fr_t *out, const Bytes32 *b);
void bytes_from_bls_field(Bytes32 *out, const fr_t *in);
void bytes_from_g1(Bytes48 *out, const g1_t *in);

#ifdef __cplusplus
}
#endif
#endif /* CKZG_H */
