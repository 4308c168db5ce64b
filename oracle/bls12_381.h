/*
 * oracle/bls12_381.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (64-bit limb, unsigned __int128) restatement of the BLS12-381 arithmetic that the
 * reference obtains from blst v0.3.16 (an un-vendored submodule: /root/reference/blst is empty;
 * pins: go.mod:7, Cargo.lock:68-71, build.zig.zon:16).  Written from the curve's mathematical
 * definition (y^2 = x^3 + 4 over Fp; tower Fp2=Fp[u]/(u^2+1), Fp6=Fp2[v]/(v^3-(1+u)),
 * Fp12=Fp6[w]/(w^2-v); optimal-ate pairing; ZCash point encoding), not from blst source.
 * The call sites it stands in for are listed per function (reference file:line).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef ORACLE_BLS12_381_H
#define ORACLE_BLS12_381_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same sizes as blst_fr / blst_fp / blst_p1 / blst_p1_affine / blst_p2
 * (bindings/go/blst_headers/blst.h:58-67,169-170,196-197): Montgomery form, little-endian limbs. */
typedef struct { uint64_t l[4]; } ofr_t;   /* element of Fr, Montgomery radix 2^256 */
typedef struct { uint64_t l[6]; } ofp_t;   /* element of Fp, Montgomery radix 2^384 */
typedef struct { ofp_t c0, c1; } ofp2_t;   /* c0 + c1*u */
typedef struct { ofp2_t c0, c1, c2; } ofp6_t;
typedef struct { ofp6_t c0, c1; } ofp12_t;
typedef struct { ofp_t x, y, z; } og1_t;        /* Jacobian; infinity <=> z == 0 */
typedef struct { ofp_t x, y; } og1_affine_t;    /* infinity encoded as (0,0) */
typedef struct { ofp2_t x, y, z; } og2_t;       /* Jacobian over Fp2 */
typedef struct { ofp2_t x, y; } og2_affine_t;

extern const ofr_t OFR_ZERO, OFR_ONE;
extern const ofp_t OFP_ZERO, OFP_ONE;
extern const og1_t OG1_IDENTITY, OG1_GENERATOR;
extern const og2_t OG2_GENERATOR;

/* ---- Fr: stands in for blst_fr_* as wrapped by src/common/fr.c:32-161 ---- */
void ofr_add(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_sub(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_neg(ofr_t *r, const ofr_t *a);
void ofr_mul(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_sqr(ofr_t *r, const ofr_t *a);
void ofr_inv(ofr_t *r, const ofr_t *a);  /* 0 -> 0, as blst_fr_eucl_inverse */
void ofr_div(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_pow(ofr_t *r, const ofr_t *a, uint64_t n);
void ofr_from_u64(ofr_t *r, uint64_t n);
bool ofr_equal(const ofr_t *a, const ofr_t *b);
bool ofr_is_zero(const ofr_t *a);
bool ofr_is_one(const ofr_t *a);
/* canonical 256-bit integer (little-endian limbs) <-> Montgomery */
void ofr_from_raw(ofr_t *r, const uint64_t raw[4]); /* raw may be >= r: reduced */
void ofr_to_raw(uint64_t raw[4], const ofr_t *a);
/* src/common/bytes.c:52-70,123-127 */
bool ofr_from_bytes_canonical(ofr_t *r, const uint8_t b[32]); /* false if >= r */
void ofr_from_bytes_reduce(ofr_t *r, const uint8_t b[32]);    /* hash_to_bls_field */
void ofr_to_bytes(uint8_t b[32], const ofr_t *a);

/* ---- Fp ---- */
void ofp_add(ofp_t *r, const ofp_t *a, const ofp_t *b);
void ofp_sub(ofp_t *r, const ofp_t *a, const ofp_t *b);
void ofp_neg(ofp_t *r, const ofp_t *a);
void ofp_mul(ofp_t *r, const ofp_t *a, const ofp_t *b);
void ofp_sqr(ofp_t *r, const ofp_t *a);
void ofp_inv(ofp_t *r, const ofp_t *a);
bool ofp_sqrt(ofp_t *r, const ofp_t *a); /* false if a is a non-residue */
bool ofp_is_zero(const ofp_t *a);
bool ofp_equal(const ofp_t *a, const ofp_t *b);
void ofp_to_raw(uint64_t raw[6], const ofp_t *a);
void ofp_from_raw(ofp_t *r, const uint64_t raw[6]);
bool ofp_from_bytes(ofp_t *r, const uint8_t b[48]); /* big-endian; false if >= p */
void ofp_to_bytes(uint8_t b[48], const ofp_t *a);
bool ofp_is_lex_largest(const ofp_t *a); /* a > (p-1)/2 */

/* ---- G1: stands in for blst_p1_* (src/common/ec.c:29-57, src/common/bytes.c:42-44,81-95) ---- */
bool og1_is_inf(const og1_t *p);
void og1_dbl(og1_t *r, const og1_t *p);
void og1_add(og1_t *r, const og1_t *a, const og1_t *b); /* complete: add-or-double */
void og1_add_affine(og1_t *r, const og1_t *a, const og1_affine_t *b);
void og1_neg(og1_t *r, const og1_t *a);
void og1_sub(og1_t *r, const og1_t *a, const og1_t *b);
void og1_mul(og1_t *r, const og1_t *p, const ofr_t *k);
void og1_mul_raw(og1_t *r, const og1_t *p, const uint64_t *k, int nbits);
bool og1_equal(const og1_t *a, const og1_t *b);
void og1_from_affine(og1_t *r, const og1_affine_t *a);
void og1_to_affine(og1_affine_t *r, const og1_t *p);
void og1_batch_to_affine(og1_affine_t *r, const og1_t *p, size_t n); /* blst_p1s_to_affine */
void og1_compress(uint8_t out[48], const og1_t *p);
/* 0 ok, 1 bad encoding, 2 not on curve -- no subgroup check (blst_p1_uncompress) */
int og1_uncompress(og1_affine_t *r, const uint8_t in[48]);
bool og1_in_subgroup(const og1_t *p);
bool og1_affine_is_inf(const og1_affine_t *a);
/* Pippenger bucket MSM, the algorithm behind blst_p1s_mult_pippenger (src/common/lincomb.c:114) */
void og1_msm_pippenger(og1_t *r, const og1_affine_t *pts, const uint64_t (*scalars)[4], size_t n,
                       int nbits);

/* ---- G2 (src/eip4844/eip4844.c:114-130, src/setup/setup.c:467-477) ---- */
bool og2_is_inf(const og2_t *p);
void og2_dbl(og2_t *r, const og2_t *p);
void og2_add(og2_t *r, const og2_t *a, const og2_t *b);
void og2_neg(og2_t *r, const og2_t *a);
void og2_mul(og2_t *r, const og2_t *p, const ofr_t *k);
void og2_to_affine(og2_affine_t *r, const og2_t *p);
void og2_from_affine(og2_t *r, const og2_affine_t *a);
int og2_uncompress(og2_affine_t *r, const uint8_t in[96]);
void og2_compress(uint8_t out[96], const og2_t *p);

/* ---- pairing check: e(a1,a2) == e(b1,b2)  (src/common/utils.c:172-196) ---- */
bool opairings_verify(const og1_t *a1, const og2_t *a2, const og1_t *b1, const og2_t *b2);

/* ---- SHA-256 (blst_sha256; src/eip4844/eip4844.c:176) ---- */
void osha256(uint8_t out[32], const uint8_t *msg, size_t len);

#ifdef __cplusplus
}
#endif
#endif
