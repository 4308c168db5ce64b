// Instruction-rate microbenchmark for the integer/FP64 ops an Fp multiplier can be built from.
// Prints, per op, giga wave-instructions/s and derived cycles per wave-instruction per SIMD
// (clock assumed from hipDeviceProp).  Run: hipcc --offload-arch=gfx950 -O3 instr_rates.hip -o ir && ./ir
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 8;   // independent chains per thread
constexpr int INNER = 64;   // ops per chain per loop trip

#define KERNEL64(NAME, ASM)                                                                  \
    __global__ void NAME(uint64_t *out, uint32_t a, uint32_t b, int iters) {                 \
        uint64_t acc[UNROLL];                                                                \
        uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                   \
        for (int u = 0; u < UNROLL; u++) acc[u] = u + threadIdx.x;                           \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int k = 0; k < INNER; k++) {                              \
                _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                         \
                    asm volatile(ASM : "+v"(acc[u]) : "v"(x), "v"(y) : "vcc");               \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
        uint64_t s = 0;                                                                      \
        for (int u = 0; u < UNROLL; u++) s ^= acc[u];                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }

#define KERNEL32(NAME, ASM)                                                                  \
    __global__ void NAME(uint64_t *out, uint32_t a, uint32_t b, int iters) {                 \
        uint32_t acc[UNROLL];                                                                \
        uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                   \
        for (int u = 0; u < UNROLL; u++) acc[u] = u + threadIdx.x;                           \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int k = 0; k < INNER; k++) {                              \
                _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                         \
                    asm volatile(ASM : "+v"(acc[u]) : "v"(x), "v"(y) : "vcc");               \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
        uint32_t s = 0;                                                                      \
        for (int u = 0; u < UNROLL; u++) s ^= acc[u];                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }

KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %0")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %0")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")

typedef void (*kern_t)(uint64_t *, uint32_t, uint32_t, int);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double clk = prop.clockRate * 1e3;  // Hz
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int threads = 256, blocks = cus * 8;
    uint64_t *out;
    CHECK(hipMalloc(&out, sizeof(uint64_t) * threads * blocks));
    struct { const char *name; kern_t k; int ops_per; } tests[] = {
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
        {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
        {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
        {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1},
        {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_add_co_u32", k_add_co_u32, 1},
        {"v_addc_co_u32", k_addc_co_u32, 1}, {"v_add_u32", k_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1},
        {"v_mov_b32", k_mov_b32, 1}, {"v_mad_u32_u16", k_mad_u32_u16, 1}, {"v_fma_f32", k_fma_f32, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &t : tests) {
        int iters = 200;
        t.k<<<blocks, threads>>>(out, 3, 5, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        t.k<<<blocks, threads>>>(out, 3, 5, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double waves = (double)blocks * threads / 64;
        double winstr = waves * iters * INNER * UNROLL * t.ops_per;
        double per_s = winstr / (ms * 1e-3);
        double cyc_per_winstr_simd = clk * cus * 4 / per_s;
        printf("%-22s %8.3f ms  %9.2f G wave-instr/s  %6.2f cyc/wave-instr/SIMD  (%7.2f T lane-ops/s)\n",
               t.name, ms, per_s / 1e9, cyc_per_winstr_simd, per_s * 64 / 1e12);
    }
    return 0;
}
