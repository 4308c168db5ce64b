/*
 * oracle/okzg.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see okzg.h).
 *
 * CPU restatement of the c-kzg-4844 v2.1.8 algorithms on and around the G1-MSM / Fr-FFT hot path.
 * All citations are file:line under /root/reference/src.
 */
#include "okzg.h"
#include <inttypes.h>
#include <stdlib.h>
#include <string.h>

#define N_G1 4096
#define N_G2 65
#define CHK(expr)                \
    do {                         \
        ret = (expr);            \
        if (ret != OKZG_OK) goto out; \
    } while (0)

static void *xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

static void be64(uint8_t out[8], uint64_t v) { /* common/bytes.c:30-35 */
    for (int i = 7; i >= 0; i--) {
        out[i] = (uint8_t)v;
        v >>= 8;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* utilities: common/utils.c                                                                    */
/* ------------------------------------------------------------------------------------------ */

static unsigned ilog2(size_t n) {
    unsigned k = 0;
    while (n >>= 1) k++;
    return k;
}

static size_t bitrev(size_t v, unsigned bits) { /* utils.c:65-88 */
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) {
        r = (r << 1) | (v & 1);
        v >>= 1;
    }
    return r;
}

int okzg_bit_reversal_permutation(void *values, size_t size, size_t n) { /* utils.c:103-140 */
    if (n < 2) return OKZG_OK;
    if (n & (n - 1)) return OKZG_BADARGS;
    uint8_t *v = values, *tmp = malloc(size);
    if (!tmp) return OKZG_MALLOC;
    unsigned bits = ilog2(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, bits);
        if (j > i) {
            memcpy(tmp, v + i * size, size);
            memcpy(v + i * size, v + j * size, size);
            memcpy(v + j * size, tmp, size);
        }
    }
    free(tmp);
    return OKZG_OK;
}

static void compute_powers(ofr_t *out, const ofr_t *x, size_t n) { /* utils.c:151-157 */
    ofr_t cur = OFR_ONE;
    for (size_t i = 0; i < n; i++) {
        out[i] = cur;
        ofr_mul(&cur, &cur, x);
    }
}

/* eip4844.c:80-106; out and a must not alias; fails on a zero input */
static int fr_batch_inv(ofr_t *out, const ofr_t *a, size_t len) {
    ofr_t acc = OFR_ONE;
    for (size_t i = 0; i < len; i++) {
        out[i] = acc;
        ofr_mul(&acc, &acc, &a[i]);
    }
    if (ofr_is_zero(&acc)) return OKZG_BADARGS;
    ofr_inv(&acc, &acc);
    for (size_t i = len; i-- > 0;) {
        ofr_mul(&out[i], &out[i], &acc);
        ofr_mul(&acc, &acc, &a[i]);
    }
    return OKZG_OK;
}

/* common/bytes.c:81-95: uncompress, accept infinity, subgroup check otherwise */
static int validate_kzg_g1(og1_t *out, const uint8_t b[48]) {
    og1_affine_t a;
    if (og1_uncompress(&a, b) != 0) return OKZG_BADARGS;
    og1_from_affine(out, &a);
    if (og1_is_inf(out)) return OKZG_OK;
    return og1_in_subgroup(out) ? OKZG_OK : OKZG_BADARGS;
}

int okzg_blob_to_polynomial(ofr_t *p, const uint8_t *blob) { /* eip4844/blob.c:31-38 */
    for (size_t i = 0; i < OKZG_FE_PER_BLOB; i++) {
        if (!ofr_from_bytes_canonical(&p[i], blob + 32 * i)) return OKZG_BADARGS;
    }
    return OKZG_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* linear combinations: common/lincomb.c                                                        */
/* ------------------------------------------------------------------------------------------ */

void okzg_g1_lincomb_naive(og1_t *out, const og1_t *p, const ofr_t *coeffs, size_t len) {
    og1_t acc = OG1_IDENTITY, t; /* lincomb.c:34-41 */
    for (size_t i = 0; i < len; i++) {
        og1_mul(&t, &p[i], &coeffs[i]);
        og1_add(&acc, &acc, &t);
    }
    *out = acc;
}

/* lincomb.c:65-123: scalars out of Montgomery form, drop points at infinity, normalise the rest
 * to affine in one batch, Pippenger over 255-bit scalars. */
int okzg_g1_lincomb_fast(og1_t *out, const og1_t *p, const ofr_t *coeffs, size_t len) {
    if (len == 0) return OKZG_BADARGS; /* c_kzg_calloc(.., 0, ..) -> BADARGS, alloc.c:52-58 */
    int ret = OKZG_OK;
    og1_t *kept = xcalloc(len, sizeof *kept);
    og1_affine_t *aff = xcalloc(len, sizeof *aff);
    uint64_t(*sc)[4] = xcalloc(len, sizeof *sc);
    if (!kept || !aff || !sc) {
        ret = OKZG_MALLOC;
        goto out;
    }
    size_t m = 0;
    for (size_t i = 0; i < len; i++) {
        if (og1_is_inf(&p[i])) continue;
        kept[m] = p[i];
        ofr_to_raw(sc[m], &coeffs[i]);
        m++;
    }
    if (m == 0) {
        *out = OG1_IDENTITY;
        goto out;
    }
    og1_batch_to_affine(aff, kept, m);
    og1_msm_pippenger(out, aff, (const uint64_t(*)[4])sc, m, 255);
out:
    free(kept);
    free(aff);
    free(sc);
    return ret;
}

/* Fixed-base windowed MSM over a precomputed table -- the role blst_p1s_mult_wbits plays at
 * fk20.c:231-239.  table[i*half + j] = (j+1)*P_i for j < half = 2^(wbits-1); scalars are recoded
 * into signed wbits-wide digits. */
static void fixed_base_msm(og1_t *out, const og1_affine_t *table, size_t wbits, size_t npoints,
                           const ofr_t *coeffs) {
    size_t half = (size_t)1 << (wbits - 1);
    int nwin = (int)((255 + wbits) / wbits); /* one spare window absorbs the last carry */
    int16_t(*dig)[260] = xcalloc(npoints, sizeof *dig);
    for (size_t i = 0; i < npoints; i++) {
        uint64_t raw[4];
        ofr_to_raw(raw, &coeffs[i]);
        int carry = 0;
        for (int w = 0; w < nwin; w++) {
            size_t lo = (size_t)w * wbits;
            uint64_t d = 0;
            if (lo < 256) {
                d = raw[lo / 64] >> (lo % 64);
                if (lo % 64 + wbits > 64 && lo / 64 + 1 < 4) d |= raw[lo / 64 + 1] << (64 - lo % 64);
                d &= ((uint64_t)1 << wbits) - 1;
            }
            int v = (int)d + carry;
            carry = 0;
            if ((size_t)v > half) {
                v -= (int)(1u << wbits);
                carry = 1;
            }
            dig[i][w] = (int16_t)v;
        }
    }
    og1_t acc = OG1_IDENTITY;
    for (int w = nwin - 1; w >= 0; w--) {
        for (size_t k = 0; k < wbits; k++) og1_dbl(&acc, &acc);
        for (size_t i = 0; i < npoints; i++) {
            int v = dig[i][w];
            if (v == 0) continue;
            og1_affine_t e = table[i * half + (size_t)(v < 0 ? -v : v) - 1];
            if (v < 0) ofp_neg(&e.y, &e.y);
            og1_add_affine(&acc, &acc, &e);
        }
    }
    free(dig);
    *out = acc;
}

/* ------------------------------------------------------------------------------------------ */
/* FFTs: eip7594/fft.c, eip7594/poly.c                                                          */
/* ------------------------------------------------------------------------------------------ */

/* fft.c:70-86: decimation in time; natural-order in, natural-order out */
static void fr_fft_rec(ofr_t *out, const ofr_t *in, size_t stride, const ofr_t *roots,
                       size_t rstride, size_t n) {
    if (n == 1) {
        *out = *in;
        return;
    }
    size_t h = n / 2;
    fr_fft_rec(out, in, stride * 2, roots, rstride * 2, h);
    fr_fft_rec(out + h, in + stride, stride * 2, roots, rstride * 2, h);
    for (size_t i = 0; i < h; i++) {
        ofr_t t;
        ofr_mul(&t, &out[i + h], &roots[i * rstride]);
        ofr_sub(&out[i + h], &out[i], &t);
        ofr_add(&out[i], &out[i], &t);
    }
}

static int fft_size_ok(size_t n) { return n <= OKZG_FE_PER_EXT_BLOB && (n & (n - 1)) == 0; }

int okzg_fr_fft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s) { /* fft.c:100 */
    if (n == 0) return OKZG_OK;
    if (!fft_size_ok(n)) return OKZG_BADARGS;
    fr_fft_rec(out, in, 1, s->roots_of_unity, OKZG_FE_PER_EXT_BLOB / n, n);
    return OKZG_OK;
}

int okzg_fr_ifft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s) { /* fft.c:127 */
    if (n == 0) return OKZG_OK;
    if (!fft_size_ok(n)) return OKZG_BADARGS;
    fr_fft_rec(out, in, 1, s->reverse_roots_of_unity, OKZG_FE_PER_EXT_BLOB / n, n);
    ofr_t inv_n;
    ofr_from_u64(&inv_n, n);
    ofr_inv(&inv_n, &inv_n);
    for (size_t i = 0; i < n; i++) ofr_mul(&out[i], &out[i], &inv_n);
    return OKZG_OK;
}

/* fft.c:164-185: the same recursion over G1; a root equal to one skips the scalar multiplication */
static void g1_fft_rec(og1_t *out, const og1_t *in, size_t stride, const ofr_t *roots,
                       size_t rstride, size_t n) {
    if (n == 1) {
        *out = *in;
        return;
    }
    size_t h = n / 2;
    g1_fft_rec(out, in, stride * 2, roots, rstride * 2, h);
    g1_fft_rec(out + h, in + stride, stride * 2, roots, rstride * 2, h);
    for (size_t i = 0; i < h; i++) {
        og1_t t;
        if (ofr_is_one(&roots[i * rstride])) {
            t = out[i + h];
        } else {
            og1_mul(&t, &out[i + h], &roots[i * rstride]);
        }
        og1_sub(&out[i + h], &out[i], &t);
        og1_add(&out[i], &out[i], &t);
    }
}

int okzg_g1_fft(og1_t *out, const og1_t *in, size_t n, const OKZGSettings *s) { /* fft.c:199 */
    if (n == 0) return OKZG_OK;
    if (!fft_size_ok(n)) return OKZG_BADARGS;
    g1_fft_rec(out, in, 1, s->roots_of_unity, OKZG_FE_PER_EXT_BLOB / n, n);
    return OKZG_OK;
}

int okzg_g1_ifft_unscaled(og1_t *out, const og1_t *in, size_t n, const OKZGSettings *s) {
    if (n == 0) return OKZG_OK; /* fft.c:227 */
    if (!fft_size_ok(n)) return OKZG_BADARGS;
    g1_fft_rec(out, in, 1, s->reverse_roots_of_unity, OKZG_FE_PER_EXT_BLOB / n, n);
    return OKZG_OK;
}

static const ofr_t SHIFT_SEVEN = {{0x0000000efffffff1ULL, 0x17e363d300189c0fULL,
                                   0xff9c57876f8457b0ULL, 0x351332208fc5a8c4ULL}};
static const ofr_t SHIFT_SEVEN_INV = {{0xdb6db6dadb6db6dcULL, 0xe6b5824adb6cc6daULL,
                                       0xf8b356e005810db9ULL, 0x66d0f1e660ec4796ULL}};

static void shift_poly(ofr_t *p, size_t len, const ofr_t *k) { /* poly.c:38-44 */
    ofr_t pw = OFR_ONE;
    for (size_t i = 1; i < len; i++) {
        ofr_mul(&pw, &pw, k);
        ofr_mul(&p[i], &p[i], &pw);
    }
}

int okzg_coset_fft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s) { /* fft.c:257 */
    if (n == 0) return OKZG_OK;
    ofr_t *tmp = malloc(n * sizeof *tmp);
    if (!tmp) return OKZG_MALLOC;
    memcpy(tmp, in, n * sizeof *tmp);
    shift_poly(tmp, n, &SHIFT_SEVEN);
    int ret = okzg_fr_fft(out, tmp, n, s);
    free(tmp);
    return ret;
}

int okzg_coset_ifft(ofr_t *out, const ofr_t *in, size_t n, const OKZGSettings *s) { /* fft.c:290 */
    if (n == 0) return OKZG_OK;
    int ret = okzg_fr_ifft(out, in, n, s);
    if (ret == OKZG_OK) shift_poly(out, n, &SHIFT_SEVEN_INV);
    return ret;
}

int okzg_poly_lagrange_to_monomial(ofr_t *out, const ofr_t *lagrange, size_t len,
                                   const OKZGSettings *s) { /* poly.c:58-80 */
    ofr_t *tmp = malloc(len * sizeof *tmp);
    if (!tmp) return OKZG_MALLOC;
    memcpy(tmp, lagrange, len * sizeof *tmp);
    int ret = okzg_bit_reversal_permutation(tmp, sizeof *tmp, len);
    if (ret == OKZG_OK) ret = okzg_fr_ifft(out, tmp, len, s);
    free(tmp);
    return ret;
}

/* ------------------------------------------------------------------------------------------ */
/* trusted setup: setup/setup.c                                                                 */
/* ------------------------------------------------------------------------------------------ */

static const ofr_t ROOT_8192 = {{0xa33d279ff0ccffc9ULL, 0x41fac79f59e91972ULL,
                                 0x065d227fead1139bULL, 0x71db41abda03e055ULL}};

static int compute_roots_of_unity(OKZGSettings *s) { /* setup.c:99-153 */
    const size_t w = OKZG_FE_PER_EXT_BLOB;
    s->roots_of_unity[0] = OFR_ONE;
    s->roots_of_unity[1] = ROOT_8192;
    size_t i;
    for (i = 2; i <= w; i++) {
        ofr_mul(&s->roots_of_unity[i], &s->roots_of_unity[i - 1], &ROOT_8192);
        if (ofr_is_one(&s->roots_of_unity[i])) break;
    }
    if (i != w || !ofr_is_one(&s->roots_of_unity[w])) return OKZG_BADARGS;
    memcpy(s->brp_roots_of_unity, s->roots_of_unity, w * sizeof(ofr_t));
    int ret = okzg_bit_reversal_permutation(s->brp_roots_of_unity, sizeof(ofr_t), w);
    if (ret != OKZG_OK) return ret;
    for (i = 0; i <= w; i++) s->reverse_roots_of_unity[i] = s->roots_of_unity[w - i];
    return OKZG_OK;
}

void okzg_free_trusted_setup(OKZGSettings *s) { /* setup.c:162-190 */
    if (!s) return;
    free(s->brp_roots_of_unity);
    free(s->roots_of_unity);
    free(s->reverse_roots_of_unity);
    free(s->g1_values_monomial);
    free(s->g1_values_lagrange_brp);
    free(s->g2_values_monomial);
    if (s->x_ext_fft_columns) {
        for (size_t i = 0; i < OKZG_CELLS_PER_EXT_BLOB; i++) free(s->x_ext_fft_columns[i]);
    }
    if (s->tables) {
        for (size_t i = 0; i < OKZG_CELLS_PER_EXT_BLOB; i++) free(s->tables[i]);
    }
    free(s->x_ext_fft_columns);
    free(s->tables);
    memset(s, 0, sizeof *s);
}

/* setup.c:238-330: for each of the 64 offsets, the FFT of the zero-extended strided slice of the
 * monomial setup points, stored column-wise; optional fixed-base tables per column. */
static int init_fk20(OKZGSettings *s) {
    int ret = OKZG_OK;
    const size_t n2 = 2 * OKZG_CELLS_PER_BLOB;
    og1_t *x = xcalloc(n2, sizeof *x), *fx = xcalloc(n2, sizeof *fx);
    s->x_ext_fft_columns = xcalloc(n2, sizeof(void *));
    if (!x || !fx || !s->x_ext_fft_columns) {
        ret = OKZG_MALLOC;
        goto out;
    }
    for (size_t i = 0; i < n2; i++) {
        s->x_ext_fft_columns[i] = xcalloc(OKZG_FE_PER_CELL, sizeof(og1_t));
        if (!s->x_ext_fft_columns[i]) {
            ret = OKZG_MALLOC;
            goto out;
        }
    }
    for (size_t off = 0; off < OKZG_FE_PER_CELL; off++) {
        size_t start = OKZG_FE_PER_BLOB - OKZG_FE_PER_CELL - 1 - off;
        for (size_t i = 0; i < n2; i++) x[i] = OG1_IDENTITY;
        for (size_t i = 0; i + 1 < OKZG_CELLS_PER_BLOB; i++) {
            x[i] = s->g1_values_monomial[start - i * OKZG_FE_PER_CELL];
        }
        CHK(okzg_g1_fft(fx, x, n2, s));
        for (size_t row = 0; row < n2; row++) s->x_ext_fft_columns[row][off] = fx[row];
    }
    if (s->wbits) {
        size_t half = (size_t)1 << (s->wbits - 1);
        s->tables = xcalloc(n2, sizeof(void *));
        og1_t *jac = xcalloc(OKZG_FE_PER_CELL * half, sizeof *jac);
        if (!s->tables || !jac) {
            free(jac);
            ret = OKZG_MALLOC;
            goto out;
        }
        for (size_t c = 0; c < n2; c++) {
            s->tables[c] = xcalloc(OKZG_FE_PER_CELL * half, sizeof(og1_affine_t));
            if (!s->tables[c]) {
                free(jac);
                ret = OKZG_MALLOC;
                goto out;
            }
            for (size_t i = 0; i < OKZG_FE_PER_CELL; i++) {
                jac[i * half] = s->x_ext_fft_columns[c][i];
                for (size_t j = 1; j < half; j++) {
                    og1_add(&jac[i * half + j], &jac[i * half + j - 1], &s->x_ext_fft_columns[c][i]);
                }
            }
            og1_batch_to_affine(s->tables[c], jac, OKZG_FE_PER_CELL * half);
        }
        free(jac);
        s->scratch_size = 0;
    }
out:
    free(x);
    free(fx);
    return ret;
}

int okzg_load_trusted_setup(OKZGSettings *out, const uint8_t *g1_mono, uint64_t n_g1_mono,
                            const uint8_t *g1_lagr, uint64_t n_g1_lagr, const uint8_t *g2_mono,
                            uint64_t n_g2, uint64_t precompute) { /* setup.c:392-505 */
    int ret = OKZG_OK;
    memset(out, 0, sizeof *out);
    if (precompute > 15) return OKZG_BADARGS;
    out->wbits = precompute;
    if (n_g1_mono != N_G1 * 48 || n_g1_lagr != N_G1 * 48 || n_g2 != N_G2 * 96) return OKZG_BADARGS;
    out->brp_roots_of_unity = xcalloc(OKZG_FE_PER_EXT_BLOB, sizeof(ofr_t));
    out->roots_of_unity = xcalloc(OKZG_FE_PER_EXT_BLOB + 1, sizeof(ofr_t));
    out->reverse_roots_of_unity = xcalloc(OKZG_FE_PER_EXT_BLOB + 1, sizeof(ofr_t));
    out->g1_values_monomial = xcalloc(N_G1, sizeof(og1_t));
    out->g1_values_lagrange_brp = xcalloc(N_G1, sizeof(og1_t));
    out->g2_values_monomial = xcalloc(N_G2, sizeof(og2_t));
    if (!out->brp_roots_of_unity || !out->roots_of_unity || !out->reverse_roots_of_unity ||
        !out->g1_values_monomial || !out->g1_values_lagrange_brp || !out->g2_values_monomial) {
        ret = OKZG_MALLOC;
        goto out;
    }
    /* the file is trusted: on-curve check only, no subgroup check (setup.c:447-477) */
    for (size_t i = 0; i < N_G1; i++) {
        og1_affine_t a;
        if (og1_uncompress(&a, g1_mono + 48 * i) != 0) {
            ret = OKZG_BADARGS;
            goto out;
        }
        og1_from_affine(&out->g1_values_monomial[i], &a);
        if (og1_uncompress(&a, g1_lagr + 48 * i) != 0) {
            ret = OKZG_BADARGS;
            goto out;
        }
        og1_from_affine(&out->g1_values_lagrange_brp[i], &a);
    }
    for (size_t i = 0; i < N_G2; i++) {
        og2_affine_t a;
        if (og2_uncompress(&a, g2_mono + 96 * i) != 0) {
            ret = OKZG_BADARGS;
            goto out;
        }
        og2_from_affine(&out->g2_values_monomial[i], &a);
    }
    /* setup.c:339-358: if e(L1, [1]_2) == e(L0, [s]_2) the "Lagrange" points are really the
     * monomial ones */
    if (opairings_verify(&out->g1_values_lagrange_brp[1], &out->g2_values_monomial[0],
                         &out->g1_values_lagrange_brp[0], &out->g2_values_monomial[1])) {
        ret = OKZG_BADARGS;
        goto out;
    }
    CHK(compute_roots_of_unity(out));
    CHK(okzg_bit_reversal_permutation(out->g1_values_lagrange_brp, sizeof(og1_t), N_G1));
    CHK(init_fk20(out));
out:
    if (ret != OKZG_OK) okzg_free_trusted_setup(out);
    return ret;
}

static int read_hex_bytes(FILE *in, uint8_t *dst, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (fscanf(in, "%2hhx", &dst[i]) != 1) return OKZG_BADARGS;
    }
    return OKZG_OK;
}

/* setup.c:519-600: "<n_g1> <n_g2>" then hex: G1 Lagrange, G2 monomial, G1 monomial */
int okzg_load_trusted_setup_file(OKZGSettings *out, FILE *in, uint64_t precompute) {
    int ret = OKZG_OK;
    uint64_t n1 = 0, n2 = 0;
    memset(out, 0, sizeof *out);
    uint8_t *mono = xcalloc(N_G1, 48), *lagr = xcalloc(N_G1, 48), *g2 = xcalloc(N_G2, 96);
    if (!mono || !lagr || !g2) {
        ret = OKZG_MALLOC;
        goto out;
    }
    if (fscanf(in, "%" SCNu64, &n1) != 1 || n1 != N_G1 || fscanf(in, "%" SCNu64, &n2) != 1 ||
        n2 != N_G2) {
        ret = OKZG_BADARGS;
        goto out;
    }
    CHK(read_hex_bytes(in, lagr, N_G1 * 48));
    CHK(read_hex_bytes(in, g2, N_G2 * 96));
    CHK(read_hex_bytes(in, mono, N_G1 * 48));
    ret = okzg_load_trusted_setup(out, mono, N_G1 * 48, lagr, N_G1 * 48, g2, N_G2 * 96, precompute);
out:
    free(mono);
    free(lagr);
    free(g2);
    return ret;
}

OKZGSettings *okzg_settings_new(void) { return calloc(1, sizeof(OKZGSettings)); }
void okzg_settings_delete(OKZGSettings *s) {
    okzg_free_trusted_setup(s);
    free(s);
}

/* ------------------------------------------------------------------------------------------ */
/* EIP-4844: eip4844/eip4844.c                                                                  */
/* ------------------------------------------------------------------------------------------ */

void okzg_compute_challenge(ofr_t *out, const uint8_t *blob, const og1_t *commitment) {
    /* eip4844.c:147-178: "FSBLOBVERIFY_V1_" | u64be 0 | u64be 4096 | blob | commitment */
    size_t len = 16 + 16 + OKZG_BYTES_PER_BLOB + 48;
    uint8_t *buf = malloc(len), h[32];
    memcpy(buf, "FSBLOBVERIFY_V1_", 16);
    be64(buf + 16, 0);
    be64(buf + 24, OKZG_FE_PER_BLOB);
    memcpy(buf + 32, blob, OKZG_BYTES_PER_BLOB);
    og1_compress(buf + 32 + OKZG_BYTES_PER_BLOB, commitment);
    osha256(h, buf, len);
    ofr_from_bytes_reduce(out, h);
    free(buf);
}

int okzg_evaluate_polynomial_in_evaluation_form(ofr_t *out, const ofr_t *poly, const ofr_t *x,
                                                const OKZGSettings *s) { /* eip4844.c:192-240 */
    int ret = OKZG_OK;
    const size_t n = OKZG_FE_PER_BLOB;
    const ofr_t *dom = s->brp_roots_of_unity;
    ofr_t *den = xcalloc(n, sizeof *den), *inv = xcalloc(n, sizeof *inv);
    if (!den || !inv) {
        ret = OKZG_MALLOC;
        goto out;
    }
    for (size_t i = 0; i < n; i++) {
        if (ofr_equal(x, &dom[i])) { /* x is in the domain: the value is stored */
            *out = poly[i];
            goto out;
        }
        ofr_sub(&den[i], x, &dom[i]);
    }
    CHK(fr_batch_inv(inv, den, n));
    ofr_t acc = OFR_ZERO, t;
    for (size_t i = 0; i < n; i++) {
        ofr_mul(&t, &inv[i], &dom[i]);
        ofr_mul(&t, &t, &poly[i]);
        ofr_add(&acc, &acc, &t);
    }
    ofr_from_u64(&t, n);
    ofr_div(&acc, &acc, &t);
    ofr_pow(&t, x, n);
    ofr_sub(&t, &t, &OFR_ONE);
    ofr_mul(out, &acc, &t);
out:
    free(den);
    free(inv);
    return ret;
}

int okzg_blob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const OKZGSettings *s) {
    int ret; /* eip4844.c:253-280 */
    og1_t c;
    ofr_t *poly = xcalloc(OKZG_FE_PER_BLOB, sizeof *poly);
    if (!poly) return OKZG_MALLOC;
    CHK(okzg_blob_to_polynomial(poly, blob));
    CHK(okzg_g1_lincomb_fast(&c, s->g1_values_lagrange_brp, poly, OKZG_FE_PER_BLOB));
    og1_compress(out, &c);
out:
    free(poly);
    return ret;
}

static int verify_kzg_proof_impl(bool *ok, const og1_t *commitment, const ofr_t *z, const ofr_t *y,
                                 const og1_t *proof, const OKZGSettings *s) {
    /* eip4844.c:359-383:  e(C - [y]G1, G2) == e(proof, [s]G2 - [z]G2) */
    og2_t zg2, nzg2, x_minus_z;
    og1_t yg1, p_minus_y;
    og2_mul(&zg2, &OG2_GENERATOR, z);
    og2_neg(&nzg2, &zg2);
    og2_add(&x_minus_z, &s->g2_values_monomial[1], &nzg2);
    og1_mul(&yg1, &OG1_GENERATOR, y);
    og1_sub(&p_minus_y, commitment, &yg1);
    *ok = opairings_verify(&p_minus_y, &OG2_GENERATOR, proof, &x_minus_z);
    return OKZG_OK;
}

int okzg_verify_kzg_proof(bool *ok, const uint8_t commitment[48], const uint8_t zb[32],
                          const uint8_t yb[32], const uint8_t proof[48], const OKZGSettings *s) {
    og1_t c, p; /* eip4844.c:313-341 */
    ofr_t z, y;
    *ok = false;
    if (validate_kzg_g1(&c, commitment) != OKZG_OK) return OKZG_BADARGS;
    if (!ofr_from_bytes_canonical(&z, zb)) return OKZG_BADARGS;
    if (!ofr_from_bytes_canonical(&y, yb)) return OKZG_BADARGS;
    if (validate_kzg_g1(&p, proof) != OKZG_OK) return OKZG_BADARGS;
    return verify_kzg_proof_impl(ok, &c, &z, &y, &p, s);
}

/* eip4844.c:417-494: y = p(z); quotient q_i = (p_i - y)/(w_i - z) in evaluation form, with the
 * special column when z is itself a domain point; proof = commit(q). */
static int compute_kzg_proof_impl(uint8_t proof_out[48], ofr_t *y_out, const ofr_t *poly,
                                  const ofr_t *z, const OKZGSettings *s) {
    int ret;
    const size_t n = OKZG_FE_PER_BLOB;
    const ofr_t *dom = s->brp_roots_of_unity;
    ofr_t *den = xcalloc(n, sizeof *den), *inv = xcalloc(n, sizeof *inv), *q = xcalloc(n, sizeof *q);
    ofr_t t;
    if (!den || !inv || !q) {
        ret = OKZG_MALLOC;
        goto out;
    }
    CHK(okzg_evaluate_polynomial_in_evaluation_form(y_out, poly, z, s));
    size_t m = 0; /* 1 + index of the domain point equal to z, if any */
    for (size_t i = 0; i < n; i++) {
        if (ofr_equal(z, &dom[i])) {
            m = i + 1;
            den[i] = OFR_ONE;
            continue;
        }
        ofr_sub(&q[i], &poly[i], y_out);
        ofr_sub(&den[i], &dom[i], z);
    }
    CHK(fr_batch_inv(inv, den, n));
    for (size_t i = 0; i < n; i++) ofr_mul(&q[i], &q[i], &inv[i]);
    if (m != 0) {
        m--;
        q[m] = OFR_ZERO;
        for (size_t i = 0; i < n; i++) {
            if (i == m) continue;
            ofr_sub(&t, z, &dom[i]);
            ofr_mul(&den[i], &t, z);
        }
        CHK(fr_batch_inv(inv, den, n));
        for (size_t i = 0; i < n; i++) {
            if (i == m) continue;
            ofr_sub(&t, &poly[i], y_out);
            ofr_mul(&t, &t, &dom[i]);
            ofr_mul(&t, &t, &inv[i]);
            ofr_add(&q[m], &q[m], &t);
        }
    }
    og1_t pr;
    CHK(okzg_g1_lincomb_fast(&pr, s->g1_values_lagrange_brp, q, n));
    og1_compress(proof_out, &pr);
out:
    free(den);
    free(inv);
    free(q);
    return ret;
}

int okzg_compute_kzg_proof(uint8_t proof_out[48], uint8_t y_out[32], const uint8_t *blob,
                           const uint8_t zb[32], const OKZGSettings *s) { /* eip4844.c:385-415 */
    int ret;
    ofr_t z, y;
    ofr_t *poly = xcalloc(OKZG_FE_PER_BLOB, sizeof *poly);
    if (!poly) return OKZG_MALLOC;
    CHK(okzg_blob_to_polynomial(poly, blob));
    if (!ofr_from_bytes_canonical(&z, zb)) {
        ret = OKZG_BADARGS;
        goto out;
    }
    CHK(compute_kzg_proof_impl(proof_out, &y, poly, &z, s));
    ofr_to_bytes(y_out, &y);
out:
    free(poly);
    return ret;
}

int okzg_compute_blob_kzg_proof(uint8_t out[48], const uint8_t *blob, const uint8_t commitment[48],
                                const OKZGSettings *s) { /* eip4844.c:496-535 */
    int ret;
    og1_t c;
    ofr_t z, y;
    ofr_t *poly = xcalloc(OKZG_FE_PER_BLOB, sizeof *poly);
    if (!poly) return OKZG_MALLOC;
    CHK(validate_kzg_g1(&c, commitment));
    CHK(okzg_blob_to_polynomial(poly, blob));
    okzg_compute_challenge(&z, blob, &c);
    CHK(compute_kzg_proof_impl(out, &y, poly, &z, s));
out:
    free(poly);
    return ret;
}

int okzg_verify_blob_kzg_proof(bool *ok, const uint8_t *blob, const uint8_t commitment[48],
                               const uint8_t proof[48], const OKZGSettings *s) {
    int ret; /* eip4844.c:537-595 */
    og1_t c, p;
    ofr_t z, y;
    *ok = false;
    ofr_t *poly = xcalloc(OKZG_FE_PER_BLOB, sizeof *poly);
    if (!poly) return OKZG_MALLOC;
    CHK(validate_kzg_g1(&c, commitment));
    CHK(okzg_blob_to_polynomial(poly, blob));
    CHK(validate_kzg_g1(&p, proof));
    okzg_compute_challenge(&z, blob, &c);
    CHK(okzg_evaluate_polynomial_in_evaluation_form(&y, poly, &z, s));
    CHK(verify_kzg_proof_impl(ok, &c, &z, &y, &p, s));
out:
    free(poly);
    return ret;
}

/* eip4844.c:597-680: r = H("RCKZGBATCH___V1_" | u64be 4096 | u64be n | (C_i|z_i|y_i|proof_i)*) */
static int compute_r_powers(ofr_t *r_powers, const og1_t *cs, const ofr_t *zs, const ofr_t *ys,
                            const og1_t *ps, size_t n) {
    size_t len = 16 + 8 + 8 + n * (48 + 32 + 32 + 48);
    uint8_t *buf = malloc(len), h[32], *o;
    if (!buf) return OKZG_MALLOC;
    memcpy(buf, "RCKZGBATCH___V1_", 16);
    be64(buf + 16, OKZG_FE_PER_BLOB);
    be64(buf + 24, n);
    o = buf + 32;
    for (size_t i = 0; i < n; i++) {
        og1_compress(o, &cs[i]);
        ofr_to_bytes(o + 48, &zs[i]);
        ofr_to_bytes(o + 80, &ys[i]);
        og1_compress(o + 112, &ps[i]);
        o += 160;
    }
    osha256(h, buf, len);
    ofr_t r;
    ofr_from_bytes_reduce(&r, h);
    compute_powers(r_powers, &r, n);
    free(buf);
    return OKZG_OK;
}

/* eip4844.c:697-758:  e(sum r^i proof_i, [s]G2) == e(sum r^i (C_i - [y_i]) + sum r^i z_i proof_i, G2) */
static int verify_kzg_proof_batch(bool *ok, const og1_t *cs, const ofr_t *zs, const ofr_t *ys,
                                  const og1_t *ps, size_t n, const OKZGSettings *s) {
    int ret;
    *ok = false;
    ofr_t *rp = xcalloc(n, sizeof *rp), *rz = xcalloc(n, sizeof *rz);
    og1_t *cmy = xcalloc(n, sizeof *cmy);
    if (!rp || !rz || !cmy) {
        ret = OKZG_MALLOC;
        goto out;
    }
    CHK(compute_r_powers(rp, cs, zs, ys, ps, n));
    og1_t proof_lc, proof_z_lc, cmy_lc, rhs, yg;
    okzg_g1_lincomb_naive(&proof_lc, ps, rp, n);
    for (size_t i = 0; i < n; i++) {
        og1_mul(&yg, &OG1_GENERATOR, &ys[i]);
        og1_sub(&cmy[i], &cs[i], &yg);
        ofr_mul(&rz[i], &rp[i], &zs[i]);
    }
    okzg_g1_lincomb_naive(&proof_z_lc, ps, rz, n);
    okzg_g1_lincomb_naive(&cmy_lc, cmy, rp, n);
    og1_add(&rhs, &cmy_lc, &proof_z_lc);
    *ok = opairings_verify(&proof_lc, &s->g2_values_monomial[1], &rhs, &OG2_GENERATOR);
out:
    free(rp);
    free(rz);
    free(cmy);
    return ret;
}

int okzg_verify_blob_kzg_proof_batch(bool *ok, const uint8_t *blobs, const uint8_t *commitments,
                                     const uint8_t *proofs, uint64_t n, const OKZGSettings *s) {
    int ret; /* eip4844.c:775-844 */
    if (n == 0) {
        *ok = true;
        return OKZG_OK;
    }
    if (n == 1) return okzg_verify_blob_kzg_proof(ok, blobs, commitments, proofs, s);
    og1_t *cs = xcalloc(n, sizeof *cs), *ps = xcalloc(n, sizeof *ps);
    ofr_t *zs = xcalloc(n, sizeof *zs), *ys = xcalloc(n, sizeof *ys);
    ofr_t *poly = xcalloc(OKZG_FE_PER_BLOB, sizeof *poly);
    if (!cs || !ps || !zs || !ys || !poly) {
        ret = OKZG_MALLOC;
        goto out;
    }
    for (size_t i = 0; i < n; i++) {
        const uint8_t *blob = blobs + i * OKZG_BYTES_PER_BLOB;
        CHK(validate_kzg_g1(&cs[i], commitments + 48 * i));
        CHK(okzg_blob_to_polynomial(poly, blob));
        okzg_compute_challenge(&zs[i], blob, &cs[i]);
        CHK(okzg_evaluate_polynomial_in_evaluation_form(&ys[i], poly, &zs[i], s));
        CHK(validate_kzg_g1(&ps[i], proofs + 48 * i));
    }
    ret = verify_kzg_proof_batch(ok, cs, zs, ys, ps, n, s);
out:
    free(cs);
    free(ps);
    free(zs);
    free(ys);
    free(poly);
    return ret;
}

/* ------------------------------------------------------------------------------------------ */
/* EIP-7594: eip7594/fk20.c, recovery.c, eip7594.c                                              */
/* ------------------------------------------------------------------------------------------ */

/* fk20.c:55-78 */
static void circulant_coeffs_stride(ofr_t *out, const ofr_t *in, size_t offset) {
    const size_t r = OKZG_CELLS_PER_BLOB, l = OKZG_FE_PER_CELL, d = OKZG_FE_PER_BLOB - 1;
    for (size_t j = 0; j < 2 * r; j++) out[j] = OFR_ZERO;
    out[0] = in[d - offset];
    for (size_t j = 1; j < r - 1; j++) out[2 * r - j] = in[d - offset - j * l];
}

/* fk20.c:139-286 */
int okzg_compute_fk20_cell_proofs(og1_t *out, const ofr_t *poly, const OKZGSettings *s) {
    int ret = OKZG_OK;
    const size_t n2 = 2 * OKZG_CELLS_PER_BLOB, l = OKZG_FE_PER_CELL;
    ofr_t *circ = xcalloc(n2, sizeof *circ), *circ_fft = xcalloc(n2, sizeof *circ_fft);
    ofr_t *coeffs = xcalloc(n2 * l, sizeof *coeffs); /* coeffs[row*l + offset] */
    og1_t *u = xcalloc(n2, sizeof *u), *v = xcalloc(n2, sizeof *v);
    if (!circ || !circ_fft || !coeffs || !u || !v) {
        ret = OKZG_MALLOC;
        goto out;
    }
    ofr_t inv_n2;
    ofr_from_u64(&inv_n2, n2);
    ofr_inv(&inv_n2, &inv_n2);
    for (size_t i = 0; i < l; i++) {
        circulant_coeffs_stride(circ, poly, i);
        CHK(okzg_fr_fft(circ_fft, circ, n2, s));
        for (size_t j = 0; j < n2; j++) ofr_mul(&coeffs[j * l + i], &circ_fft[j], &inv_n2);
    }
    for (size_t i = 0; i < n2; i++) {
        if (s->wbits) {
            fixed_base_msm(&u[i], s->tables[i], s->wbits, l, &coeffs[i * l]);
        } else {
            CHK(okzg_g1_lincomb_fast(&u[i], s->x_ext_fft_columns[i], &coeffs[i * l], l));
        }
    }
    CHK(okzg_g1_ifft_unscaled(v, u, n2, s));
    for (size_t i = OKZG_CELLS_PER_BLOB; i < n2; i++) v[i] = OG1_IDENTITY;
    CHK(okzg_g1_fft(out, v, n2, s));
out:
    free(circ);
    free(circ_fft);
    free(coeffs);
    free(u);
    free(v);
    return ret;
}

static void cells_from_fr(uint8_t *cells, const ofr_t *data) { /* eip7594.c:113-120 */
    for (size_t i = 0; i < OKZG_FE_PER_EXT_BLOB; i++) ofr_to_bytes(cells + 32 * i, &data[i]);
}

static int proofs_to_bytes(uint8_t *proofs, og1_t *pr) { /* eip7594.c:133-147 */
    int ret = okzg_bit_reversal_permutation(pr, sizeof *pr, OKZG_CELLS_PER_EXT_BLOB);
    if (ret != OKZG_OK) return ret;
    for (size_t i = 0; i < OKZG_CELLS_PER_EXT_BLOB; i++) og1_compress(proofs + 48 * i, &pr[i]);
    return OKZG_OK;
}

int okzg_compute_cells_and_kzg_proofs(uint8_t *cells, uint8_t *proofs, const uint8_t *blob,
                                      const OKZGSettings *s) { /* eip7594.c:61-157 */
    int ret;
    if (!cells && !proofs) return OKZG_BADARGS;
    const size_t n = OKZG_FE_PER_EXT_BLOB;
    ofr_t *mono = xcalloc(n, sizeof *mono), *lagr = xcalloc(n, sizeof *lagr);
    ofr_t *data = xcalloc(n, sizeof *data);
    og1_t *pr = xcalloc(OKZG_CELLS_PER_EXT_BLOB, sizeof *pr);
    if (!mono || !lagr || !data || !pr) {
        ret = OKZG_MALLOC;
        goto out;
    }
    CHK(okzg_blob_to_polynomial(lagr, blob));
    CHK(okzg_poly_lagrange_to_monomial(mono, lagr, OKZG_FE_PER_BLOB, s)); /* upper half stays 0 */
    if (cells) {
        CHK(okzg_fr_fft(data, mono, n, s));
        CHK(okzg_bit_reversal_permutation(data, sizeof *data, n));
        cells_from_fr(cells, data);
    }
    if (proofs) {
        CHK(okzg_compute_fk20_cell_proofs(pr, mono, s));
        CHK(proofs_to_bytes(proofs, pr));
    }
out:
    free(mono);
    free(lagr);
    free(data);
    free(pr);
    return ret;
}

/* recovery.c:46-75: prod (x - root_i), coefficients low to high */
static int vanishing_poly_from_roots(ofr_t *poly, size_t *poly_len, const ofr_t *roots, size_t n) {
    if (n == 0) return OKZG_BADARGS;
    ofr_t nr;
    ofr_neg(&poly[0], &roots[0]);
    for (size_t i = 1; i < n; i++) {
        ofr_neg(&nr, &roots[i]);
        ofr_add(&poly[i], &nr, &poly[i - 1]);
        for (size_t j = i - 1; j > 0; j--) {
            ofr_mul(&poly[j], &poly[j], &nr);
            ofr_add(&poly[j], &poly[j], &poly[j - 1]);
        }
        ofr_mul(&poly[0], &poly[0], &nr);
    }
    poly[n] = OFR_ONE;
    *poly_len = n + 1;
    return OKZG_OK;
}

/* recovery.c:93-162 */
static int vanishing_poly_for_missing_cells(ofr_t *vanishing, const uint64_t *missing, size_t nm,
                                            const OKZGSettings *s) {
    int ret;
    if (nm == 0 || nm >= OKZG_CELLS_PER_EXT_BLOB) return OKZG_BADARGS;
    ofr_t *roots = xcalloc(nm, sizeof *roots), *shortp = xcalloc(nm + 1, sizeof *shortp);
    size_t short_len = 0;
    if (!roots || !shortp) {
        ret = OKZG_MALLOC;
        goto out;
    }
    size_t stride = OKZG_FE_PER_EXT_BLOB / OKZG_CELLS_PER_EXT_BLOB;
    for (size_t i = 0; i < nm; i++) roots[i] = s->roots_of_unity[missing[i] * stride];
    CHK(vanishing_poly_from_roots(shortp, &short_len, roots, nm));
    for (size_t i = 0; i < OKZG_FE_PER_EXT_BLOB; i++) vanishing[i] = OFR_ZERO;
    for (size_t i = 0; i < short_len; i++) vanishing[i * OKZG_FE_PER_CELL] = shortp[i];
out:
    free(roots);
    free(shortp);
    return ret;
}

/* recovery.c:200-365 */
int okzg_recover_cells(ofr_t *out, const uint64_t *cell_indices, size_t num_cells, ofr_t *cells,
                       const OKZGSettings *s) {
    int ret;
    const size_t n = OKZG_FE_PER_EXT_BLOB;
    uint64_t *missing = xcalloc(OKZG_CELLS_PER_EXT_BLOB, sizeof *missing);
    ofr_t *zeval = xcalloc(n, sizeof(ofr_t)), *zcoef = xcalloc(n, sizeof(ofr_t));
    ofr_t *ez = xcalloc(n, sizeof(ofr_t)), *ezc = xcalloc(n, sizeof(ofr_t));
    ofr_t *eoc = xcalloc(n, sizeof(ofr_t)), *zoc = xcalloc(n, sizeof(ofr_t));
    ofr_t *rec = xcalloc(n, sizeof(ofr_t)), *brp = xcalloc(n, sizeof(ofr_t));
    if (!missing || !zeval || !zcoef || !ez || !ezc || !eoc || !zoc || !rec || !brp) {
        ret = OKZG_MALLOC;
        goto out;
    }
    memcpy(brp, cells, n * sizeof(ofr_t));
    CHK(okzg_bit_reversal_permutation(brp, sizeof(ofr_t), n));
    size_t nm = 0;
    for (size_t i = 0; i < OKZG_CELLS_PER_EXT_BLOB; i++) {
        bool have = false;
        for (size_t k = 0; k < num_cells; k++) have |= (cell_indices[k] == i);
        if (!have) missing[nm++] = bitrev(i, 7);
    }
    CHK(vanishing_poly_for_missing_cells(zcoef, missing, nm, s));
    CHK(okzg_fr_fft(zeval, zcoef, n, s));
    for (size_t i = 0; i < n; i++) ofr_mul(&ez[i], &brp[i], &zeval[i]);
    CHK(okzg_fr_ifft(ezc, ez, n, s));
    CHK(okzg_coset_fft(eoc, ezc, n, s));
    CHK(okzg_coset_fft(zoc, zcoef, n, s));
    for (size_t i = 0; i < n; i++) ofr_div(&eoc[i], &eoc[i], &zoc[i]);
    CHK(okzg_coset_ifft(rec, eoc, n, s));
    CHK(okzg_fr_fft(out, rec, n, s));
    CHK(okzg_bit_reversal_permutation(out, sizeof(ofr_t), n));
out:
    free(missing);
    free(zeval);
    free(zcoef);
    free(ez);
    free(ezc);
    free(eoc);
    free(zoc);
    free(rec);
    free(brp);
    return ret;
}

int okzg_recover_cells_and_kzg_proofs(uint8_t *recovered_cells, uint8_t *recovered_proofs,
                                      const uint64_t *cell_indices, const uint8_t *cells,
                                      uint64_t num_cells, const OKZGSettings *s) {
    int ret; /* eip7594.c:177-304 */
    const size_t n = OKZG_FE_PER_EXT_BLOB;
    if (num_cells > OKZG_CELLS_PER_EXT_BLOB || num_cells < OKZG_CELLS_PER_BLOB) return OKZG_BADARGS;
    for (size_t i = 0; i < num_cells; i++) {
        if (cell_indices[i] >= OKZG_CELLS_PER_EXT_BLOB) return OKZG_BADARGS;
        if (i > 0 && cell_indices[i] <= cell_indices[i - 1]) return OKZG_BADARGS;
    }
    ofr_t *data = xcalloc(n, sizeof *data);
    og1_t *pr = xcalloc(OKZG_CELLS_PER_EXT_BLOB, sizeof *pr);
    if (!data || !pr) {
        ret = OKZG_MALLOC;
        goto out;
    }
    for (size_t i = 0; i < num_cells; i++) {
        size_t base = cell_indices[i] * OKZG_FE_PER_CELL;
        for (size_t j = 0; j < OKZG_FE_PER_CELL; j++) {
            if (!ofr_from_bytes_canonical(&data[base + j], cells + i * OKZG_BYTES_PER_CELL + 32 * j)) {
                ret = OKZG_BADARGS;
                goto out;
            }
        }
    }
    if (num_cells == OKZG_CELLS_PER_EXT_BLOB) {
        memcpy(recovered_cells, cells, OKZG_CELLS_PER_EXT_BLOB * OKZG_BYTES_PER_CELL);
    } else {
        CHK(okzg_recover_cells(data, cell_indices, num_cells, data, s));
        cells_from_fr(recovered_cells, data);
    }
    if (recovered_proofs) {
        CHK(okzg_poly_lagrange_to_monomial(data, data, n, s));
        CHK(okzg_compute_fk20_cell_proofs(pr, data, s));
        CHK(proofs_to_bytes(recovered_proofs, pr));
    }
out:
    free(data);
    free(pr);
    return ret;
}

/* eip7594.c:390-482 */
int okzg_compute_verify_cell_kzg_proof_batch_challenge(
    ofr_t *out, const uint8_t *commitments, uint64_t num_commitments,
    const uint64_t *commitment_indices, const uint64_t *cell_indices, const uint8_t *cells,
    const uint8_t *proofs, uint64_t num_cells) {
    size_t len = 16 + 4 * 8 + num_commitments * 48 + num_cells * (8 + 8 + OKZG_BYTES_PER_CELL + 48);
    uint8_t *buf = malloc(len), h[32], *o;
    if (!buf) return OKZG_MALLOC;
    memcpy(buf, "RCKZGCBATCH__V1_", 16);
    be64(buf + 16, OKZG_FE_PER_BLOB);
    be64(buf + 24, OKZG_FE_PER_CELL);
    be64(buf + 32, num_commitments);
    be64(buf + 40, num_cells);
    o = buf + 48;
    memcpy(o, commitments, num_commitments * 48);
    o += num_commitments * 48;
    for (size_t i = 0; i < num_cells; i++) {
        be64(o, commitment_indices[i]);
        be64(o + 8, cell_indices[i]);
        memcpy(o + 16, cells + i * OKZG_BYTES_PER_CELL, OKZG_BYTES_PER_CELL);
        memcpy(o + 16 + OKZG_BYTES_PER_CELL, proofs + 48 * i, 48);
        o += 16 + OKZG_BYTES_PER_CELL + 48;
    }
    osha256(h, buf, len);
    ofr_from_bytes_reduce(out, h);
    free(buf);
    return OKZG_OK;
}

/* eip7594.c:825-974 with its helpers :345-376, :494-539, :549-601, :615-772, :784-812 */
int okzg_verify_cell_kzg_proof_batch(bool *ok, const uint8_t *commitments,
                                     const uint64_t *cell_indices, const uint8_t *cells,
                                     const uint8_t *proofs, uint64_t num_cells,
                                     const OKZGSettings *s) {
    int ret;
    *ok = false;
    if (num_cells == 0) {
        *ok = true;
        return OKZG_OK;
    }
    for (size_t i = 0; i < num_cells; i++) {
        if (cell_indices[i] >= OKZG_CELLS_PER_EXT_BLOB) return OKZG_BADARGS;
    }
    const size_t n = num_cells, next = OKZG_FE_PER_EXT_BLOB, l = OKZG_FE_PER_CELL;
    uint8_t *uniq = xcalloc(n, 48);
    uint64_t *cidx = xcalloc(n, sizeof *cidx);
    ofr_t *rp = xcalloc(n, sizeof *rp), *wts = xcalloc(n, sizeof *wts), *wrp = xcalloc(n, sizeof *wrp);
    og1_t *pg = xcalloc(n, sizeof *pg), *cg = xcalloc(n, sizeof *cg);
    ofr_t *agg = xcalloc(next, sizeof *agg), *col = xcalloc(l, sizeof *col);
    ofr_t *interp = xcalloc(l, sizeof *interp);
    bool *used = xcalloc(OKZG_CELLS_PER_EXT_BLOB, sizeof *used);
    if (!uniq || !cidx || !rp || !wts || !wrp || !pg || !cg || !agg || !col || !interp || !used) {
        ret = OKZG_MALLOC;
        goto out;
    }
    /* deduplicate commitments, remembering each cell's index into the unique list */
    size_t nc = 0;
    for (size_t i = 0; i < n; i++) {
        size_t j;
        for (j = 0; j < nc; j++) {
            if (memcmp(uniq + 48 * j, commitments + 48 * i, 48) == 0) break;
        }
        if (j == nc) memcpy(uniq + 48 * nc++, commitments + 48 * i, 48);
        cidx[i] = j;
    }
    ofr_t r;
    CHK(okzg_compute_verify_cell_kzg_proof_batch_challenge(&r, uniq, nc, cidx, cell_indices, cells,
                                                          proofs, n));
    compute_powers(rp, &r, n);
    for (size_t i = 0; i < n; i++) CHK(validate_kzg_g1(&pg[i], proofs + 48 * i));
    og1_t proof_lc, csum, interp_commit, wsum;
    CHK(okzg_g1_lincomb_fast(&proof_lc, pg, rp, n));
    /* sum over unique commitments of (sum of r^i of their cells) * C */
    for (size_t j = 0; j < nc; j++) {
        CHK(validate_kzg_g1(&cg[j], uniq + 48 * j));
        wts[j] = OFR_ZERO;
    }
    for (size_t i = 0; i < n; i++) ofr_add(&wts[cidx[i]], &wts[cidx[i]], &rp[i]);
    CHK(okzg_g1_lincomb_fast(&csum, cg, wts, nc));
    /* commitment to the aggregated interpolation polynomial */
    for (size_t i = 0; i < n; i++) {
        for (size_t j = 0; j < l; j++) {
            ofr_t v;
            if (!ofr_from_bytes_canonical(&v, cells + i * OKZG_BYTES_PER_CELL + 32 * j)) {
                ret = OKZG_BADARGS;
                goto out;
            }
            ofr_mul(&v, &v, &rp[i]);
            ofr_t *dst = &agg[cell_indices[i] * l + j];
            ofr_add(dst, dst, &v);
        }
        used[cell_indices[i]] = true;
    }
    for (size_t c = 0; c < OKZG_CELLS_PER_EXT_BLOB; c++) {
        if (!used[c]) continue;
        CHK(okzg_bit_reversal_permutation(&agg[c * l], sizeof(ofr_t), l));
        CHK(okzg_fr_ifft(col, &agg[c * l], l, s));
        /* divide out the coset shift h_c = w^brp(c): multiply coefficient k by h_c^-k */
        size_t rb = bitrev(c, 7);
        shift_poly(col, l, &s->roots_of_unity[OKZG_FE_PER_EXT_BLOB - rb]);
        for (size_t k = 0; k < l; k++) ofr_add(&interp[k], &interp[k], &col[k]);
    }
    CHK(okzg_g1_lincomb_fast(&interp_commit, s->g1_values_monomial, interp, l));
    og1_sub(&csum, &csum, &interp_commit);
    /* sum r^i * h_k^64 * proof_i */
    for (size_t i = 0; i < n; i++) {
        size_t rb = bitrev(cell_indices[i], 7);
        ofr_mul(&wrp[i], &rp[i], &s->roots_of_unity[rb * l]);
    }
    CHK(okzg_g1_lincomb_fast(&wsum, pg, wrp, n));
    og1_add(&csum, &csum, &wsum);
    *ok = opairings_verify(&csum, &OG2_GENERATOR, &proof_lc, &s->g2_values_monomial[l]);
out:
    free(uniq);
    free(cidx);
    free(rp);
    free(wts);
    free(wrp);
    free(pg);
    free(cg);
    free(agg);
    free(col);
    free(interp);
    free(used);
    return ret;
}
