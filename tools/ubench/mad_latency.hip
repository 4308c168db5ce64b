// How much instruction-level parallelism does a wave need to keep v_mad_u64_u32 at its issue rate?
// NCH independent accumulator chains per lane, 1 / 2 / 8 waves per SIMD.  Also the rates of the
// bookkeeping instructions of the 28-bit-limb Montgomery product (fp28.hpp): v_mul_lo_u32,
// v_lshrrev_b64, v_and_b32, v_lshl_add_u64.  Decides between row-wise (15 accumulators) and
// column-wise (Comba, 1-2 accumulators) product scanning.
// Run: hipcc --offload-arch=gfx950 -O3 mad_latency.hip -o mad_latency && ./mad_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NCH>
__global__ void k_mad(uint64_t *out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[NCH];
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int u = 0; u < NCH; u++) acc[u] = u + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 512 / NCH; k++) {
#pragma unroll
            for (int u = 0; u < NCH; u++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(x), "v"(y) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int u = 0; u < NCH; u++) s ^= acc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define KERNEL32(NAME, ASM)                                                                  \
    __global__ void NAME(uint64_t *out, uint32_t a, uint32_t b, int iters) {                 \
        uint32_t acc[8];                                                                     \
        uint32_t x = a + threadIdx.x;                                                        \
        for (int u = 0; u < 8; u++) acc[u] = u + threadIdx.x;                                \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int k = 0; k < 64; k++) {                                 \
                _Pragma("unroll") for (int u = 0; u < 8; u++) asm volatile(ASM : "+v"(acc[u]) : "v"(x)); \
            }                                                                                \
        }                                                                                    \
        uint32_t s = 0;                                                                      \
        for (int u = 0; u < 8; u++) s ^= acc[u];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }
#define KERNEL64(NAME, ASM)                                                                  \
    __global__ void NAME(uint64_t *out, uint32_t a, uint32_t b, int iters) {                 \
        uint64_t acc[8];                                                                     \
        uint32_t x = a + threadIdx.x;                                                        \
        for (int u = 0; u < 8; u++) acc[u] = ((uint64_t)x << 33) + u;                        \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int k = 0; k < 64; k++) {                                 \
                _Pragma("unroll") for (int u = 0; u < 8; u++) asm volatile(ASM : "+v"(acc[u]) : "v"(x)); \
            }                                                                                \
        }                                                                                    \
        uint64_t s = 0;                                                                      \
        for (int u = 0; u < 8; u++) s ^= acc[u];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_lshr32, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 28")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 28")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 1, %0")
KERNEL64(k_lshl_add64, "v_lshl_add_u64 %0, %0, 0, %0")

typedef void (*kern_t)(uint64_t *, uint32_t, uint32_t, int);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double clk = prop.clockRate * 1e3;
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    uint64_t *out;
    CHECK(hipMalloc(&out, sizeof(uint64_t) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto run = [&](const char *name, kern_t k, int wps, double ops_per_iter) {
        int blocks = cus * wps, iters = 100;
        k<<<blocks, 256>>>(out, 3, 5, 5);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<<<blocks, 256>>>(out, 3, 5, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double winstr = (double)blocks * 4 * iters * ops_per_iter;
        double per_s = winstr / (ms * 1e-3);
        printf("%-28s waves/SIMD %d  %7.2f cyc/wave-instr/SIMD  cycles between dependent issues of one wave ~%6.2f\n", name, wps,
               clk * cus * 4 / per_s, clk * cus * 4 / per_s * wps);
    };
    for (int wps : {1, 2, 8}) {
        run("v_mad_u64_u32 chains=1", k_mad<1>, wps, 512);
        run("v_mad_u64_u32 chains=2", k_mad<2>, wps, 512);
        run("v_mad_u64_u32 chains=4", k_mad<4>, wps, 512);
        run("v_mad_u64_u32 chains=8", k_mad<8>, wps, 512);
        run("v_mad_u64_u32 chains=16", k_mad<16>, wps, 512);
    }
    for (int wps : {2, 8}) {
        run("v_mul_lo_u32", k_mul_lo, wps, 512);
        run("v_and_b32", k_and, wps, 512);
        run("v_add_u32", k_add, wps, 512);
        run("v_lshrrev_b32", k_lshr32, wps, 512);
        run("v_bfe_u32", k_bfe, wps, 512);
        run("v_alignbit_b32", k_alignbit, wps, 512);
        run("v_lshrrev_b64", k_lshr64, wps, 512);
        run("v_lshl_add_u64", k_lshl_add64, wps, 512);
    }
    return 0;
}
