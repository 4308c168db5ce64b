// exec_skip.hip -- does a wave64 VALU instruction get cheaper when only part of the wave is active?
// A wave64 instruction issues in four passes of 16 lanes.  k_sha256_challenges is issue-bound with ONE lane per blob
// (925 instructions per 64-byte block at ~4.8 cycles each: 4.84 ms for any batch up to 65,536 blobs, profiles/
// r05_sha_round_order_ab.txt).  If the hardware skipped the passes whose 16 lanes are all inactive, 16 blobs per wave
// on four times as many waves would hash a 4096-blob batch four times faster.  Measured here: a dependent chain of simple
// VALU instructions with 64 / 32 / 16 / 1 active lanes per wave, one wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/exec_skip.hip -o tools/ubench/exec_skip && tools/ubench/exec_skip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int NCH>
__global__ void k_chain(uint32_t *out, uint32_t a, int iters, int active) {
    uint32_t acc[NCH];
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x = a + threadIdx.x, y = a ^ (threadIdx.x * 7u);
    for (int u = 0; u < NCH; u++) acc[u] = u * 77u + threadIdx.x;
    if (lane < (uint32_t)active) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 256 / NCH; k++) {
#pragma unroll
                for (int u = 0; u < NCH; u++) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc[u]) : "v"(x), "v"(y));
            }
        }
    }
    uint32_t s = 0;
    for (int u = 0; u < NCH; u++) s ^= acc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int nch : {1, 4}) {
        for (int active : {64, 48, 32, 16, 8, 1}) {
            const int blocks = cus, iters = 4000;
            auto launch = [&](int it) {
                if (nch == 1) k_chain<1><<<blocks, 256>>>(out, 3, it, active);
                else k_chain<4><<<blocks, 256>>>(out, 3, it, active);
            };
            launch(10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("chains=%d active lanes %2d: %6.2f ns per instruction of ONE wave (= %.2f cycles at 2.1 GHz)\n", nch, active,
                   ms * 1e6 / ((double)iters * 256), ms * 1e6 / ((double)iters * 256) * 2.1);
        }
    }
    return 0;
}
