// Throughput + correctness probe for the compiler-generated 12x32-bit Montgomery multiplier.
#include "../../c-kzg-4844_amd/csrc/field.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ckzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class F>
__global__ void k_mulchain(F *out, const F *a, const F *b, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F x = a[i], y = b[i];
    for (int k = 0; k < iters; k++) { x = mul(x, y); y = mul(y, x); }
    out[i] = add(x, y);
}

template <class F, class P>
void run(const char *name, int blocks, int threads, int iters) {
    size_t n = (size_t)blocks * threads;
    std::vector<F> a(n), b(n), o(n);
    srand(1);
    for (size_t i = 0; i < n; i++) {
        uint32_t ra[P::N], rb[P::N];
        for (int k = 0; k < P::N; k++) { ra[k] = rand() * 65537u + rand(); rb[k] = rand() * 65537u + rand(); }
        ra[P::N - 1] &= 0x0fffffff; rb[P::N - 1] &= 0x0fffffff;
        a[i] = from_raw<P>(ra); b[i] = from_raw<P>(rb);
    }
    F *da, *db, *dout;
    CHECK(hipMalloc(&da, n * sizeof(F))); CHECK(hipMalloc(&db, n * sizeof(F))); CHECK(hipMalloc(&dout, n * sizeof(F)));
    CHECK(hipMemcpy(da, a.data(), n * sizeof(F), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), n * sizeof(F), hipMemcpyHostToDevice));
    k_mulchain<F><<<blocks, threads>>>(dout, da, db, 4);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    k_mulchain<F><<<blocks, threads>>>(dout, da, db, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(o.data(), dout, n * sizeof(F), hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < n; i += n / 64) {
        F x = a[i], y = b[i];
        for (int k = 0; k < iters; k++) { x = mul(x, y); y = mul(y, x); }
        F r = add(x, y);
        if (r != o[i]) bad++;
    }
    double muls = (double)n * iters * 2;
    printf("%s: blocks %d threads %d: %.3f ms, %.3e mul/s, mismatches vs host: %d\n", name, blocks, threads, ms, muls / (ms * 1e-3), bad);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    run<Fp, FpParams>("Fp mul (12 limbs)", cus * 8, 256, 2000);
    run<Fp, FpParams>("Fp mul (12 limbs)", cus * 4, 256, 2000);
    run<Fp, FpParams>("Fp mul (12 limbs)", cus * 2, 256, 2000);
    run<Fr, FrParams>("Fr mul (8 limbs)", cus * 8, 256, 4000);
    return 0;
}
