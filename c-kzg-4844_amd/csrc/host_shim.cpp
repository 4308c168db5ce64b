// host_shim.cpp -- exports the header-only host/device arithmetic through a C ABI so the CPU
// test-suite (tests/test_host_arith.py) can compare it limb for limb with the oracle.  Built with
// plain g++ (no GPU needed); it is a test aid for the product's own headers, not part of
// libckzg_hip.so.
#include "host_pairing.hpp"
using namespace ckzg;
using namespace ckzg::host;

extern "C" {
void hs_fp_mul(Fp *r, const Fp *a, const Fp *b) { *r = mul(*a, *b); }
void hs_fp_add(Fp *r, const Fp *a, const Fp *b) { *r = add(*a, *b); }
void hs_fp_sub(Fp *r, const Fp *a, const Fp *b) { *r = sub(*a, *b); }
void hs_fp_inv(Fp *r, const Fp *a) { *r = fp_inv(*a); }
void hs_fr_mul(Fr *r, const Fr *a, const Fr *b) { *r = mul(*a, *b); }
void hs_fr_add(Fr *r, const Fr *a, const Fr *b) { *r = add(*a, *b); }
void hs_fr_sub(Fr *r, const Fr *a, const Fr *b) { *r = sub(*a, *b); }
void hs_fr_inv(Fr *r, const Fr *a) { *r = fr_inv(*a); }
// G1: all in/out as Jacobian (blst_p1 layout)
void hs_g1_add_jac(G1Jac *r, const G1Jac *a, const G1Jac *b) { *r = jac_add(*a, *b); }
void hs_g1_dbl_jac(G1Jac *r, const G1Jac *a) { *r = jac_dbl(*a); }
void hs_g1_add_xyzz(G1Jac *r, const G1Jac *a, const G1Jac *b) {
    *r = jac_from_xyzz(xyzz_add(xyzz_from_jac(*a), xyzz_from_jac(*b)));
}
void hs_g1_madd_xyzz(G1Jac *r, const G1Jac *a, const G1Affine *b) {
    G1XYZZ acc = xyzz_from_jac(*a);
    xyzz_madd(acc, *b);
    *r = jac_from_xyzz(acc);
}
void hs_g1_madd_jac(G1Jac *r, const G1Jac *a, const G1Affine *b) { *r = jac_madd(*a, *b); }
void hs_g1_dbl_xyzz(G1Jac *r, const G1Jac *a) { *r = jac_from_xyzz(xyzz_dbl(xyzz_from_jac(*a))); }
void hs_g1_mul(G1Jac *r, const G1Jac *a, const uint32_t *k, int nbits) { *r = jac_mul(*a, k, nbits); }
void hs_g1_compress(uint8_t *out, const G1Jac *a) { g1_compress_affine(out, jac_to_affine(*a)); }
int hs_g1_uncompress(G1Affine *out, const uint8_t *in) { return g1_uncompress(*out, in); }
void hs_g1_xyzz_to_affine(G1Affine *out, const G1Jac *a) { *out = xyzz_to_affine(xyzz_from_jac(*a)); }
int hs_g2_uncompress(G2Affine *out, const uint8_t *in) { return g2_uncompress(*out, in); }
int hs_pairings_verify(const G1Jac *a1, const G2Jac *a2, const G1Jac *b1, const G2Jac *b2) {
    return pairings_verify(*a1, *a2, *b1, *b2) ? 1 : 0;
}
void hs_g2_mul(G2Jac *r, const G2Jac *a, const uint32_t *k, int nbits) { *r = g2_mul(*a, k, nbits); }
void hs_g2_generator(G2Jac *r) { *r = g2_generator(); }
void hs_g1_generator(G1Jac *r) { *r = g1_generator(); }
void hs_sha256(uint8_t *out, const uint8_t *msg, size_t len) {
    Sha256 s;
    // feed in awkward pieces to exercise the buffering
    size_t a = len / 3;
    s.update(msg, a);
    s.update(msg + a, len - a);
    s.finish(out);
}
}
