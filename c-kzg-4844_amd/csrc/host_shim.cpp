// host_shim.cpp -- exports the header-only host/device arithmetic through a C ABI so the CPU
// test-suite (tests/test_host_arith.py) can compare it limb for limb with the oracle.  Built with
// plain g++ (no GPU needed); it is a test aid for the product's own headers, not part of
// libckzg_hip.so.
#include <vector>
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
// portable compression loop only (the dispatching Sha256 picks SHA-NI when the CPU has it)
void hs_sha256_portable(uint8_t *out, const uint8_t *msg, size_t len) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t nb = len / 64;
    for (size_t i = 0; i < nb; i++) sha256_block(h, msg + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * nb;
    memcpy(tail, msg + 64 * nb, rem);
    tail[rem] = 0x80;
    size_t tl = rem < 56 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (size_t i = 0; i < tl / 64; i++) sha256_block(h, tail + 64 * i);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
}
int hs_cpu_has_sha_ni() {
#ifdef CKZG_HAVE_SHANI
    return cpu_has_sha_ni() ? 1 : 0;
#else
    return 0;
#endif
}
void hs_sha256(uint8_t *out, const uint8_t *msg, size_t len) {
    Sha256 s;
    // feed in awkward pieces to exercise the buffering
    size_t a = len / 3;
    s.update(msg, a);
    s.update(msg + a, len - a);
    s.finish(out);
}
}

// ---- 28-bit-limb device arithmetic (fp28.hpp / g1_28.hpp), exercised on the host ----
#include "g1_28.hpp"
#include "fr_inv.hpp"
extern "C" {
// r = a*b in the 2^384 domain, computed through the 2^392-domain multiplier
void hs_fp28_mul(Fp *r, const Fp *a, const Fp *b) { *r = f28_to_fp(mul(f28_from_fp(*a), f28_from_fp(*b))); }
void hs_fp28_roundtrip(Fp *r, const Fp *a) { *r = f28_to_fp(f28_from_fp(*a)); }
void hs_fp28_sub(Fp *r, const Fp *a, const Fp *b) { *r = f28_to_fp(sub(f28_from_fp(*a), f28_from_fp(*b))); }
void hs_fp28_addchain(Fp *r, const Fp *a, const Fp *b) {
    auto x = f28_from_fp(*a), y = f28_from_fp(*b);
    *r = f28_to_fp(norm(sub(add(add(x, y), x), add(y, y))));  // 2a - b
}
// r = (-a)*b through cneg_reduced (a must be fully reduced, as table coordinates are)
void hs_fp28_cneg_mul(Fp *r, const Fp *a, const Fp *b, int negate) {
    Fp k;
    for (int i = 0; i < 12; i++) k.l[i] = FP_MONT_2POW8[i];
    Fp t = mul(*a, k);  // a in the 2^392 domain, fully reduced, packed
    *r = f28_to_fp(mul(cneg_reduced(f28_unpack<1>(t.l), negate != 0), f28_from_fp(*b)));
}
// table entries are stored as the 32-bit-domain product with 2^8, packed; emulate that path
static F28<1, 1> table_coord(const Fp &v) {
    Fp k;
    for (int i = 0; i < 12; i++) k.l[i] = FP_MONT_2POW8[i];
    Fp t = mul(v, k);
    return f28_unpack<1>(t.l);
}
// acc (Jacobian, may be infinity) += +-pt (affine) through xyzz28_madd
void hs_g1_madd28(G1Jac *r, const G1Jac *acc_in, const G1Affine *pt, int negate) {
    XYZZ28 acc;
    bool inf = acc_in->is_inf();
    if (!inf) {
        G1XYZZ a = xyzz_from_jac(*acc_in);
        acc.x = widen<1, 10>(f28_from_fp(a.x));
        acc.y = widen<1, 6>(f28_from_fp(a.y));
        acc.zz = f28_from_fp(a.zz);
        acc.zzz = f28_from_fp(a.zzz);
    }
    xyzz28_madd(acc, inf, table_coord(pt->x), cneg_reduced(table_coord(pt->y), negate != 0));
    *r = jac_from_xyzz(xyzz28_to_xyzz(acc, inf));
}
void hs_g1_add28(G1Jac *r, const G1Jac *a, const G1Jac *b) {
    bool ai, bi;
    XYZZ28 x = xyzz28_from_xyzz(xyzz_from_jac(*a), ai), y = xyzz28_from_xyzz(xyzz_from_jac(*b), bi);
    xyzz28_add(x, ai, y, bi);
    *r = jac_from_affine(xyzz28_to_affine(x, ai));
}
void hs_g1_mul28(G1Jac *r, const G1Jac *a, const uint32_t *k) {
    bool ai, oi;
    XYZZ28 x = xyzz28_from_xyzz(xyzz_from_jac(*a), ai), o;
    xyzz28_mul_w4(o, oi, x, ai, k);
    *r = jac_from_affine(xyzz28_to_affine(o, oi));
}
// subgroup test and square root of the device validation path; returns 1 in / 0 out / 2 not on curve
int hs_g1_in_subgroup28(const G1Affine *pt) {
    auto x = f28_from_fp(pt->x), y = f28_from_fp(pt->y);
    F28<1, 2> yy;
    if (!g1_28_solve_y(yy, x)) return 2;
    if (!f28_equal(yy, y) && !is_zero(mul(add(yy, y), f28_one()))) return 3;  // sqrt must be +-y
    return g1_28_in_subgroup(x, y) ? 1 : 0;
}
void hs_glv_split(uint32_t *k1k2, const uint32_t *k) { glv_split(k, k1k2, k1k2 + 4); }
void hs_g1_mul28_glv(G1Jac *r, const G1Jac *a, const uint32_t *k) {
    uint32_t glv[8];
    glv_split(k, glv, glv + 4);
    bool ai, oi;
    XYZZ28 x = xyzz28_from_xyzz(xyzz_from_jac(*a), ai), o;
    xyzz28_mul_glv_w4(o, oi, x, ai, glv);
    *r = jac_from_affine(xyzz28_to_affine(o, oi));
}
void hs_wnaf4_128(int8_t *out, const uint32_t *k) { wnaf4_128(out, k); }
void hs_g1_mul28_glv_naf(G1Jac *r, const G1Jac *a, const uint32_t *k) {
    uint32_t glv[8];
    glv_split(k, glv, glv + 4);
    int8_t naf[2 * GLV_NAF_LEN];
    wnaf4_128(naf, glv);
    wnaf4_128(naf + GLV_NAF_LEN, glv + 4);
    bool ai, oi;
    XYZZ28 x = xyzz28_from_xyzz(xyzz_from_jac(*a), ai), o;
    xyzz28_mul_glv_naf(o, oi, x, ai, naf, naf + GLV_NAF_LEN);
    *r = jac_from_affine(xyzz28_to_affine(o, oi));
}
// The co-Z table of the NAF ladder and its mixed addition, step by step: r[m] = tbl[m] brought home (m = 0..3:
// P, 3P, 5P, 7P); then with acc on the table's curve: r[4] = inf + P, r[5] = P + P (the equal-points branch),
// r[6] = 2P + 3P (generic), r[7] = 5P + (-5P) (must be infinity), r[8] = 2P + phi-entry of 7P.
void hs_je28_cases(G1Jac *r, const G1Jac *a) {
    bool ai;
    XYZZ28 x = xyzz28_from_xyzz(xyzz_from_jac(*a), ai);
    EAT28 tbl[4];
    F28<1, 2> zc;
    eat28_build(tbl, zc, x);
    auto home = [&](const JE28 &acc, bool inf) {
        if (inf) return jac_from_affine(xyzz28_to_affine(x, true));
        JAC28 j;
        j.x = widen<1, 34>(acc.x);
        j.y = widen<1, 34>(acc.y);
        j.z = widen<2, 4>(mul(acc.z, zc));
        return jac_from_affine(xyzz28_to_affine(jac28_to_xyzz(j), false));
    };
    for (int m = 0; m < 4; m++) {
        JE28 acc;
        bool inf = true;
        je28_madd(acc, inf, tbl[m].x, tbl[m].y, false);
        r[m] = home(acc, inf);
    }
    JE28 acc;
    bool inf = true;
    je28_madd(acc, inf, tbl[0].x, tbl[0].y, false);
    r[4] = home(acc, inf);
    je28_madd(acc, inf, tbl[0].x, tbl[0].y, false);
    r[5] = home(acc, inf);
    JE28 two = acc;
    je28_madd(acc, inf, tbl[1].x, tbl[1].y, false);
    r[6] = home(acc, inf);
    je28_madd(acc, inf, tbl[2].x, tbl[2].y, true);
    r[7] = home(acc, inf);
    acc = two;
    inf = false;
    je28_madd(acc, inf, tbl[3].bx, tbl[3].y, false);
    r[8] = home(acc, inf);
}
// a + (+-b) and 2a through the Jacobian 28-bit-limb formulas (b must be finite)
void hs_g1_jac28_add(G1Jac *r, const G1Jac *a, const G1Jac *b, int negate_b) {
    bool ai, bi;
    XYZZ28 xa = xyzz28_from_xyzz(xyzz_from_jac(*a), ai), xb = xyzz28_from_xyzz(xyzz_from_jac(*b), bi);
    JAC28 ja;
    if (!ai) ja = jac28_from_xyzz(xa);
    JACT28 tb = jac28_table_entry(jac28_from_xyzz(xb));
    if (negate_b) tb = jact28_neg(tb);
    jac28_add(ja, ai, tb);
    XYZZ28 o;
    if (!ai) o = jac28_to_xyzz(ja);
    *r = jac_from_affine(xyzz28_to_affine(o, ai));
}
void hs_g1_jac28_dbl_chain(G1Jac *r, const G1Jac *a, int n) {
    bool ai;
    XYZZ28 xa = xyzz28_from_xyzz(xyzz_from_jac(*a), ai);
    JAC28 ja = jac28_from_xyzz(xa);
    for (int i = 0; i < n; i++) jac28_dbl(ja);
    *r = jac_from_affine(xyzz28_to_affine(jac28_to_xyzz(ja), false));
}
void hs_g1_neg28(G1Jac *r, const G1Jac *a) {
    bool ai;
    XYZZ28 x = xyzz28_from_xyzz(xyzz_from_jac(*a), ai);
    *r = jac_from_affine(xyzz28_to_affine(xyzz28_neg(x), ai));
}
// r = a*b + c*d with one reduction, operands lazily reduced the way xyzz28_madd_alt feeds them:
// (a + a) as <2,4>, (b - c) as <4,18>
void hs_fp28_mul_add2(Fp *r, const Fp *a, const Fp *b, const Fp *c, const Fp *d) {
    auto fa = f28_from_fp(*a), fb = f28_from_fp(*b), fc = f28_from_fp(*c), fd = f28_from_fp(*d);
    auto a2 = add(fa, fa);                              // <2,4>
    auto bc = sub(fb, widen<1, 10>(fc));                // <4,18>
    *r = f28_to_fp(mul_add2(a2, bc, widen<1, 6>(fc), fd));  // 2a(b-c) + c*d
}
// chain through xyzz28_madd_alt (sign-alternating accumulator): signs[i] != 0 subtracts pts[i]
void hs_g1_madd28_alt_chain(G1Jac *r, const G1Affine *pts, const uint8_t *signs, int n) {
    XYZZ28 acc;
    bool inf = true, yneg = false;
    for (int i = 0; i < n; i++)
        xyzz28_madd_alt(acc, inf, yneg, table_coord(pts[i].x), table_coord(pts[i].y), signs[i] != 0);
    xyzz28_fix_sign(acc, inf, yneg);
    *r = jac_from_xyzz(xyzz28_to_xyzz(acc, inf));
}
// many additions in a row, to exercise the value-bound bookkeeping over a long chain
void hs_g1_madd28_chain(G1Jac *r, const G1Affine *pts, int n) {
    XYZZ28 acc;
    bool inf = true;
    for (int i = 0; i < n; i++) xyzz28_madd(acc, inf, table_coord(pts[i].x), cneg_reduced(table_coord(pts[i].y), (i & 1) != 0));
    *r = jac_from_xyzz(xyzz28_to_xyzz(acc, inf));
}
}

// Consistency of the fast Fp12 routines with the generic product, on pseudo-random inputs derived
// from `seed`: bit 0 complex squaring, bit 1 sparse line product, bit 2 cyclotomic squaring (on an
// element of the cyclotomic subgroup), bit 3 cyclotomic squaring must DIFFER on a generic element
// (guards against a test that cannot fail).  Returns 0 when everything agrees.
extern "C" int hs_g1_in_subgroup_host(const G1Affine *a) { return g1_in_subgroup_host(*a) ? 1 : 0; }
extern "C" void hs_g1_mul_glv_host(G1Jac *r, const G1Jac *p, const uint32_t *k) { *r = g1_mul_glv_host(*p, k); }

extern "C" void hs_fr_inv_safegcd(Fr *r, const Fr *a) { *r = fr_inv_safegcd(*a); }
extern "C" void hs_fr_inv_fermat(Fr *r, const Fr *a) { *r = fr_inv(*a); }

// timing helpers (tools only): n pairing-product checks with prepared G2 arguments / n Fp products
extern "C" int hs_bench_pairing(const G1Jac *a1, const G2Jac *q1, const G1Jac *a2, const G2Jac *q2, int n) {
    G2Prepared p1, p2;
    g2_prepare(p1, g2_to_affine(*q1));
    g2_prepare(p2, g2_to_affine(*q2));
    G1Affine x1 = jac_to_affine(*a1), x2 = jac_to_affine(*a2);
    int ok = 0;
    for (int i = 0; i < n; i++) ok += pairing_product_is_one(x1, p1, x2, p2) ? 1 : 0;
    return ok;
}
// n final exponentiations of a fixed Miller-loop value (timing only)
extern "C" int hs_bench_final_exp(const G1Jac *a1, const G2Jac *q1, int n) {
    Fp12 f = miller_loop(g2_to_affine(*q1), jac_to_affine(*a1));
    int ones = 0;
    for (int i = 0; i < n; i++) ones += final_exp(f).is_one() ? 1 : 0;
    return ones;
}
extern "C" void hs_bench_fp_mul(Fp *r, const Fp *a, const Fp *b, int n) {
    Fp x = *a;
    for (int i = 0; i < n; i++) x = mul(x, *b);
    *r = x;
}

extern "C" int hs_fp12_selftest(uint32_t seed) {
    auto rnd_fp = [&](uint32_t i) {
        uint8_t in[8], d[64];
        memcpy(in, &seed, 4);
        memcpy(in + 4, &i, 4);
        Sha256 a;
        a.update(in, 8);
        a.finish(d);
        Sha256 b;
        b.update(d, 32);
        b.finish(d + 32);
        uint32_t raw[12];
        memcpy(raw, d, 48);
        raw[11] &= 0x0fffffffu;  // < p
        return from_raw<FpParams>(raw);
    };
    uint32_t ctr = 0;
    auto rnd_fp2 = [&]() { Fp2 r = {rnd_fp(ctr), rnd_fp(ctr + 1)}; ctr += 2; return r; };
    auto rnd_fp12 = [&]() {
        Fp12 f;
        f.c0 = {rnd_fp2(), rnd_fp2(), rnd_fp2()};
        f.c1 = {rnd_fp2(), rnd_fp2(), rnd_fp2()};
        return f;
    };
    auto same = [](const Fp12 &a, const Fp12 &b) { return std::memcmp(&a, &b, sizeof a) == 0; };
    int bad = 0;
    Fp12 f = rnd_fp12();
    if (!same(sqr(f), mul(f, f))) bad |= 1;
    {
        Fp2 lam = rnd_fp2(), c = rnd_fp2();
        G1Affine p = {rnd_fp(ctr), rnd_fp(ctr + 1)};
        ctr += 2;
        Fp12 l;
        std::memset(&l, 0, sizeof l);
        l.c0.c0 = c;
        l.c0.c1 = neg(mul_fp(lam, p.x));
        l.c1.c1.c0 = p.y;
        if (!same(mul_by_prepared_line(f, lam, c, p), mul(f, l))) bad |= 2;
    }
    Fp12 a = mul(conj(f), inv(f));
    a = mul(frobenius(a, 2), a);  // in the cyclotomic subgroup
    if (!same(cyclotomic_sqr(a), mul(a, a))) bad |= 4;
    if (same(cyclotomic_sqr(f), mul(f, f))) bad |= 8;
    // bit 4: the lazily reduced Fp2 product and square against the schoolbook formulas on Fp
    for (int rep = 0; rep < 8; rep++) {
        Fp2 x = rnd_fp2(), y = rnd_fp2();
        if (rep == 0) y = x;
        if (rep == 1) { x.c0 = Fp::zero(); y.c1 = Fp::zero(); }
        if (rep == 2) { x.c0 = neg(Fp::one()); x.c1 = neg(Fp::one()); y = x; }  // p - 1 in both slots
        Fp2 want = {sub(mul(x.c0, y.c0), mul(x.c1, y.c1)), add(mul(x.c0, y.c1), mul(x.c1, y.c0))};
        Fp2 got = mul(x, y);
        if (!(got == want)) bad |= 16;
        Fp2 sq = sqr(x), sqw = {sub(mul(x.c0, x.c0), mul(x.c1, x.c1)), dbl(mul(x.c0, x.c1))};
        if (!(sq == sqw)) bad |= 16;
    }
    return bad;
}

// [k]G1 from the generator table against the generic ladder; returns 1 when they agree
extern "C" int hs_g1_gen_mul_check(const uint32_t *k) {
    G1Jac a = g1_gen_mul(k), b = g1_mul_glv_host(g1_generator(), k);
    G1Affine x = jac_to_affine(a), y = jac_to_affine(b);
    return std::memcmp(&x, &y, sizeof x) == 0 ? 1 : 0;
}
// the product check with the second pair's Miller loop run separately (what verify_blob_kzg_proof does underneath
// the GPU's evaluation) against the fused loop: returns verdict | (agreement << 1)
extern "C" int hs_pairing_split(const G1Jac *a1, const G2Jac *q1, const G1Jac *a2, const G2Jac *q2) {
    G2Prepared p1, p2;
    g2_prepare(p1, g2_to_affine(*q1));
    g2_prepare(p2, g2_to_affine(*q2));
    const G1Affine x1 = jac_to_affine(*a1), x2 = jac_to_affine(*a2);
    const Fp12 early = miller_product_prepared(x2, p2, G1Affine::inf(), p2);
    const Fp12 late = miller_product_prepared(x1, p1, G1Affine::inf(), p1);
    const bool split = final_exp(mul(late, early)).is_one();
    const bool fused = pairing_product_is_one(x1, p1, x2, p2);
    return (split ? 1 : 0) | ((split == fused) ? 2 : 0);
}

extern "C" int hs_pairing_prepared(const G1Jac *a1, const G2Jac *q1, const G1Jac *a2, const G2Jac *q2) {
    G2Prepared p1, p2;
    g2_prepare(p1, g2_to_affine(*q1));
    g2_prepare(p2, g2_to_affine(*q2));
    return pairing_product_is_one(jac_to_affine(*a1), p1, jac_to_affine(*a2), p2) ? 1 : 0;
}

extern "C" {
void hs_fp28_sqr(Fp *r, const Fp *a) { *r = f28_to_fp(sqr(f28_from_fp(*a))); }
// square of a lazily reduced operand (limbs up to 4 units, value up to 18p): (a - b)^2
void hs_fp28_sqr_lazy(Fp *r, const Fp *a, const Fp *b) {
    auto d = sub(f28_from_fp(*a), widen<1, 10>(f28_from_fp(*b)));  // <4,18>
    *r = f28_to_fp(sqr(d));
}
void hs_fp28_inv(Fp *r, const Fp *a) { *r = f28_to_fp(f28_inv_fermat(f28_from_fp(*a))); }
}

extern "C" void hs_fp28_inv_safegcd(Fp *r, const Fp *a) { *r = f28_to_fp(f28_inv_safegcd(f28_from_fp(*a))); }

// balanced GLV split + signed window recoding of the fixed-base MSM kernels (g1_28.hpp)
extern "C" void hs_glv_split_signed(const uint32_t *k, uint32_t *m1, int *neg1, uint32_t *m2, int *neg2) {
    bool n1, n2;
    glv_split_signed(k, m1, n1, m2, n2);
    *neg1 = n1;
    *neg2 = n2;
}
extern "C" void hs_recode_signed_128(int16_t *dst, const uint32_t *m, int neg, int wbits, int nwh) {
    recode_signed_128(dst, 1, m, neg != 0, wbits, nwh);
}

// The fixed-base MSM kernels' algorithm (msm.hip: k_msm_accumulate / k_msm_small) replayed on the host
// with the very same inline functions: GLV digits, table entries (mag * 2^(wbits*tw) * P_i, computed here
// on demand instead of read from HBM), sign-alternating mixed additions, phi applied once per lane when it
// crosses from the k2 half into the k1 half, lane partials folded with xyzz28_add.
extern "C" void hs_msm_glv_emulate(G1Jac *r, const G1Affine *pts, const uint32_t *scalars, int n, int wbits,
                                   int lanes) {
    const int twin = 127 / wbits + 1, nwin = 2 * twin;
    std::vector<int16_t> dg((size_t)nwin * n);
    for (int i = 0; i < n; i++) {
        uint32_t m1[4], m2[4];
        bool n1, n2;
        glv_split_signed(scalars + 8 * i, m1, n1, m2, n2);
        recode_signed_128(dg.data() + i, n, m2, n2, wbits, twin);
        recode_signed_128(dg.data() + (size_t)twin * n + i, n, m1, n1, wbits, twin);
    }
    const uint32_t pairs = (uint32_t)nwin * n, phi_pairs = pairs / 2;
    XYZZ28 total;
    bool tinf = true;
    for (int l = 0; l < lanes; l++) {
        XYZZ28 acc;
        bool inf = true, yneg = false, phi_pending = (uint32_t)l < phi_pairs;
        for (uint32_t q = l; q < pairs; q += lanes) {
            if (phi_pending && q >= phi_pairs) {
                if (!inf) acc.x = widen<1, 10>(mul(acc.x, f28_const<1, 1>(FP28_BETA_LAMBDA)));
                phi_pending = false;
            }
            int d = dg[q];
            if (d == 0) continue;
            uint32_t w = q / n, i = q - w * n, tw = w >= (uint32_t)twin ? w - twin : w;
            uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            G1Jac e = jac_from_affine(pts[i]);
            for (uint32_t k = 0; k < tw * (uint32_t)wbits; k++) e = jac_dbl(e);
            G1Jac m = G1Jac::inf();
            for (int b = 31; b >= 0; b--) {
                m = jac_dbl(m);
                if ((mag >> b) & 1u) m = jac_add(m, e);
            }
            if (m.is_inf()) continue;  // a table entry at infinity is skipped by the kernels too
            G1Affine a = jac_to_affine(m);
            xyzz28_madd_alt(acc, inf, yneg, table_coord(a.x), table_coord(a.y), d < 0);
        }
        if (phi_pending && !inf) acc.x = widen<1, 10>(mul(acc.x, f28_const<1, 1>(FP28_BETA_LAMBDA)));
        xyzz28_fix_sign(acc, inf, yneg);
        xyzz28_add(total, tinf, acc, inf);
    }
    *r = jac_from_affine(xyzz28_to_affine(total, tinf));
}

// ---- Fr on 29-bit limbs (fr29.hpp) and the barycentric evaluation built on it, replayed thread by thread ----
#include "fr29.hpp"
extern "C" void hs_fr29_mul(Fr *r, const Fr *a, const Fr *b) { *r = fr29_to_fr(fr29_mul(fr29_from_fr(*a), fr29_from_fr(*b))); }
// the radix travels with the operand: (a 2^256) (b 2^261) / 2^261 is the library's form of a b
extern "C" void hs_fr29_mul_mixed(Fr *r, const Fr *a, const Fr *b) {
    fr29_unpack(r->l, fr29_canonical<0>(fr29_mul(fr29_pack(a->l), fr29_from_fr(*b))));
}
extern "C" void hs_fr29_roundtrip(Fr *r, const Fr *a) { *r = fr29_to_fr(fr29_from_fr(*a)); }
extern "C" void hs_fr29_inv(Fr *r, const Fr *a) { *r = fr29_to_fr(fr29_inv(fr29_from_fr(*a))); }
extern "C" void hs_fr29_sub(Fr *r, const Fr *a, const Fr *b) {
    *r = fr29_to_fr(fr29_sub_canonical(fr29_from_fr(*a), fr29_from_fr(*b)));
}
// sum of n values through the lazy additions and the closing ladder (n <= 32)
extern "C" void hs_fr29_sum(Fr *r, const Fr *a, int n) {
    Fr29 s;
    for (int i = 0; i < 9; i++) s.l[i] = 0;
    for (int k = 0; k < n; k++) {
        Fr29 t = fr29_pack(a[k].l);
        for (int j = 0; j < 9; j++) s.l[j] += t.l[j];
        if ((k & 3) == 3) fr29_carry(s);
    }
    fr29_carry(s);
    *r = ev29::to_fr_radix256(s);
}
// One blob through the kernel's algorithm (verify.hip: k_eval_barycentric): 512 threads of 8 terms, one inversion
// per lane of the first wave for the eight waves' products.  Returns the index of the domain point equal to z (y =
// p[hit]) or -1; di_out (may be null): 4096 x 8 words, 1/(z - w_i) in the 2^261 radix, canonical.
extern "C" int hs_fr29_eval(Fr *y, uint32_t *di_out, const Fr *poly, const Fr *z, const Fr *brp_roots) {
    constexpr int T = 64 * ev29::WAVES, NB = 4096;
    static_assert(T * ev29::PER == NB, "geometry");
    std::vector<Fr29> roots29(NB), pre((size_t)T * ev29::PER), acc(T), parked(T), inv(T);
    for (int i = 0; i < NB; i++) roots29[i] = fr29_from_fr(brp_roots[i]);
    const Fr29 z29 = fr29_from_fr(*z);
    int hit = -1;
    for (int t = 0; t < T; t++) {
        int h = ev29::forward(&pre[(size_t)t * ev29::PER], acc[t], z29, roots29.data(), t, T);
        if (h >= 0) hit = h;
    }
    if (hit >= 0) {
        *y = poly[hit];
        return hit;
    }
    for (int l = 0; l < 64; l++)
        ev29::invert_across([&](int w) { return acc[l + 64 * w]; }, [&](int w, const Fr29 &v) { parked[l + 64 * w] = v; },
                            [&](int w) { return parked[l + 64 * w]; }, [&](int w, const Fr29 &v) { inv[l + 64 * w] = v; },
                            [](const Fr29 &v) { return fr29_inv(v); });
    Fr sum = Fr::zero();
    for (int t = 0; t < T; t++)
        sum = add(sum, ev29::to_fr_radix256(ev29::backward(&pre[(size_t)t * ev29::PER], inv[t], z29, roots29.data(), poly, t, T, di_out)));
    *y = ev29::scale(sum, ev29::vanishing_over_n(z29));
    return -1;
}

// One blob through k_eval_tree<LOG_PER>'s algorithm: 4096 >> LOG_PER threads each fold 2^LOG_PER leaves, six levels
// of lane exchanges inside a wave (both lanes of a pair compute the parent), the waves' values by thread 0.
// bytes != nullptr: the leaves come from the blob's bytes (k_eval_tree's BYTES form); *bad |= 1 for an element >= r.
template <int LOG_PER>
static void eval_tree_emulate(Fr *y, const Fr *poly, const Fr *z, const Fr *brp_roots, const uint8_t *bytes = nullptr,
                              uint32_t *bad = nullptr) {
    constexpr int NB = 4096, T = NB >> LOG_PER, W = T / 64;
    std::vector<Fr29> tab(NB / 2), v(T), nv(T);
    for (int m = 0; m < NB / 2; m++) tab[m] = fr29_inv(fr29_from_fr(brp_roots[2 * m]));
    for (auto &t : tab) t = fr29_canonical<0>(t);
    Fr29 zp[12];
    zp[0] = fr29_from_fr(*z);
    for (int l = 1; l < 12; l++) zp[l] = fr29_mul(zp[l - 1], zp[l - 1]);
    uint32_t any_bad = 0;
    auto leaf = [&](int i) {
        if (!bytes) return fr29_pack(poly[i].l);
        uint32_t s[8];
        for (int k = 0; k < 8; k++) {
            const uint8_t *q = bytes + 32 * i + 4 * (7 - k);
            s[k] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | q[3];
        }
        return ev29::tree_leaf_from_words(s, any_bad);
    };
    for (int t = 0; t < T; t++)
        v[t] = ev29::tree_canonical<LOG_PER>(ev29::tree_node<LOG_PER>(leaf, tab.data(), zp, t << LOG_PER));
    if (bad) *bad |= any_bad;
    for (int k = 0; k < 6; k++) {
        for (int t = 0; t < T; t++) {
            const bool odd = (t >> k) & 1;
            const Fr29 &mine = v[t], &other = v[t ^ (1 << k)];
            nv[t] = fr29_canonical<1>(ev29::tree_combine<0>(odd ? other : mine, odd ? mine : other,
                                                            fr29_mul(zp[LOG_PER + k], tab[t >> (k + 1)])));
        }
        v.swap(nv);
    }
    Fr29 a[W > 1 ? W : 1];
    for (int w = 0; w < W; w++) a[w] = v[64 * w];
    int lvl = LOG_PER + 6;
    for (int n = W; n > 1; n >>= 1, lvl++)
        for (int m = 0; m < n / 2; m++)
            a[m] = fr29_canonical<1>(ev29::tree_combine<0>(a[2 * m], a[2 * m + 1], fr29_mul(zp[lvl], tab[m])));
    *y = bytes ? ev29::tree_finish_from_integers(a[0]) : ev29::tree_finish(a[0]);
}
extern "C" void hs_fr29_eval_tree_bytes(Fr *y, uint32_t *bad, const uint8_t *blob, const Fr *z, const Fr *brp_roots, int log_per) {
    if (log_per == 6) eval_tree_emulate<6>(y, nullptr, z, brp_roots, blob, bad);
    else eval_tree_emulate<4>(y, nullptr, z, brp_roots, blob, bad);
}
extern "C" void hs_fr29_eval_tree(Fr *y, const Fr *poly, const Fr *z, const Fr *brp_roots, int log_per) {
    if (log_per == 6) eval_tree_emulate<6>(y, poly, z, brp_roots);
    else eval_tree_emulate<4>(y, poly, z, brp_roots);
}
