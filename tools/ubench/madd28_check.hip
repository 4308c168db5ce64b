// Device-vs-host check of the 28-bit-limb XYZZ mixed addition chain (g1_28.hpp).
#include "../../c-kzg-4844_amd/csrc/g1_28.hpp"
#include "../../c-kzg-4844_amd/csrc/host_pairing.hpp"
#include <cstdio>
#include <vector>
using namespace ckzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

__host__ __device__ inline F28<1, 1> table_coord(const Fp &v) {
    Fp k;
    for (int i = 0; i < 12; i++) k.l[i] = FP_MONT_2POW8[i];
    Fp t = mul(v, k);
    return f28_unpack<1>(t.l);
}

__global__ void k_chain(G1XYZZ *out, const G1Affine *pts, int n, int per_thread) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ28 acc;
    bool inf = true;
    for (int i = 0; i < per_thread; i++) {
        int idx = (t * 7 + i * 13) % n;
        xyzz28_madd(acc, inf, table_coord(pts[idx].x), cneg_reduced(table_coord(pts[idx].y), (i & 1) != 0));
    }
    out[t] = xyzz28_to_xyzz(acc, inf);
}

__global__ void k_mulchk(Fp *out, const Fp *a, const Fp *b, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = f28_to_fp(mul(f28_from_fp(a[t]), f28_from_fp(b[t])));
}

int main() {
    const int n = 64;
    std::vector<G1Affine> pts(n);
    G1Jac g = host::g1_generator();
    uint32_t k[8] = {0x12345, 0x9abcdef, 0x7777, 1, 2, 3, 4, 0x0fffffff};
    G1Jac cur = jac_mul(g, k, 255);
    for (int i = 0; i < n; i++) { pts[i] = jac_to_affine(cur); cur = jac_add(jac_dbl(cur), g); }
    // field mul check
    std::vector<Fp> a(n), b(n), o(n);
    for (int i = 0; i < n; i++) { a[i] = pts[i].x; b[i] = pts[i].y; }
    Fp *da, *db, *dout; CHECK(hipMalloc(&da, n * sizeof(Fp))); CHECK(hipMalloc(&db, n * sizeof(Fp))); CHECK(hipMalloc(&dout, n * sizeof(Fp)));
    CHECK(hipMemcpy(da, a.data(), n * sizeof(Fp), hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, b.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    k_mulchk<<<1, 64>>>(dout, da, db, n); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(o.data(), dout, n * sizeof(Fp), hipMemcpyDeviceToHost));
    int badm = 0; for (int i = 0; i < n; i++) if (o[i] != mul(a[i], b[i])) badm++;
    printf("fp28 mul device mismatches: %d\n", badm);
    G1Affine *dp; G1XYZZ *dres; const int threads = 128;
    CHECK(hipMalloc(&dp, n * sizeof(G1Affine))); CHECK(hipMalloc(&dres, threads * sizeof(G1XYZZ)));
    CHECK(hipMemcpy(dp, pts.data(), n * sizeof(G1Affine), hipMemcpyHostToDevice));
    for (int per : {1, 2, 3, 8, 16, 40}) {
        k_chain<<<threads / 64, 64>>>(dres, dp, n, per); CHECK(hipDeviceSynchronize());
        std::vector<G1XYZZ> res(threads);
        CHECK(hipMemcpy(res.data(), dres, threads * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
        int bad = 0, badref = 0;
        for (int t = 0; t < threads; t++) {
            XYZZ28 acc; bool inf = true; G1Jac ref = G1Jac::inf();
            for (int i = 0; i < per; i++) {
                int idx = (t * 7 + i * 13) % n;
                xyzz28_madd(acc, inf, table_coord(pts[idx].x), cneg_reduced(table_coord(pts[idx].y), (i & 1) != 0));
                G1Affine q = pts[idx]; if (i & 1) q = affine_neg(q);
                ref = jac_madd(ref, q);
            }
            G1XYZZ h = xyzz28_to_xyzz(acc, inf);
            G1Affine ha = xyzz_to_affine(h), da2 = xyzz_to_affine(res[t]), ra = jac_to_affine(ref);
            if (!(ha.x == da2.x && ha.y == da2.y)) bad++;
            if (!(ha.x == ra.x && ha.y == ra.y)) badref++;
        }
        printf("chain per_thread=%d: device!=host %d, host!=reference %d\n", per, bad, badref);
    }
    return 0;
}
