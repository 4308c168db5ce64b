// How should a handful of Fermat inversions be laid out?  (A) one 64-lane block per value with a
// single active lane, (B) one lane per value packed into full waves, (C) 256-thread blocks, one
// wave per value with a single active lane.
#include "../../c-kzg-4844_amd/csrc/g1_28.hpp"
#include <cstdio>
#include <vector>
using namespace ckzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

__global__ void kA(Fp *out, const Fp *in) {  // block per value, lane 0 works
    if (threadIdx.x == 0) out[blockIdx.x] = f28_to_fp(f28_inv(f28_from_fp(in[blockIdx.x])));
}
__global__ void kB(Fp *out, const Fp *in, int n) {  // lane per value
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f28_to_fp(f28_inv(f28_from_fp(in[i])));
}
__global__ void kC(Fp *out, const Fp *in, int n) {  // wave per value, 4 waves per block
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && i < n) out[i] = f28_to_fp(f28_inv(f28_from_fp(in[i])));
}
__global__ void kOld(Fp *out, const Fp *in, int n) {  // lane per value, 32-bit-limb code
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fp_inv(in[i]);
}

int main() {
    const int n = 1024;
    std::vector<Fp> h(n), o(n);
    for (int i = 0; i < n; i++) { h[i] = Fp::one(); h[i].l[0] += i * 7 + 3; h[i].l[5] ^= i * 0x9e37; h[i].l[11] &= 0x0fffffff; }
    Fp *din, *dout; CHECK(hipMalloc(&din, n * sizeof(Fp))); CHECK(hipMalloc(&dout, n * sizeof(Fp)));
    CHECK(hipMemcpy(din, h.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; variant++) {
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (variant == 0) kA<<<n, 64>>>(dout, din);
            if (variant == 1) kB<<<n / 64, 64>>>(dout, din, n);
            if (variant == 2) kC<<<n / 4, 256>>>(dout, din, n);
            if (variant == 3) kOld<<<n / 64, 64>>>(dout, din, n);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) {
                CHECK(hipMemcpy(o.data(), dout, n * sizeof(Fp), hipMemcpyDeviceToHost));
                int bad = 0; for (int i = 0; i < n; i += 37) if (mul(o[i], h[i]) != Fp::one()) bad++;
                printf("variant %c: %.3f ms for %d inversions (bad %d)\n", "ABCO"[variant], ms, n, bad);
            }
        }
    }
    return 0;
}
