// valu_dep_issue.hip -- what does a LONE wave pay for back-to-back DEPENDENT simple VALU instructions?
// k_sha256_challenges is one wave per SIMD walking a chain of 925 instructions per block; it measures 4.83 cycles per
// instruction where 4.0 is the issue rate of a wave64 instruction.  If a dependent pair costs more than an independent
// one, the round's instruction ORDER (not its count) has slack.  NCH independent chains per lane, one wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_dep_issue.hip -o tools/ubench/valu_dep_issue && tools/ubench/valu_dep_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define KERNEL(NAME, NCH, ASM)                                                                       \
    __global__ void NAME(uint32_t *out, uint32_t a, int iters) {                                     \
        uint32_t acc[NCH];                                                                           \
        uint32_t x = a + threadIdx.x, y = a ^ (threadIdx.x * 7u);                                    \
        for (int u = 0; u < NCH; u++) acc[u] = u * 77u + threadIdx.x;                                \
        for (int it = 0; it < iters; it++) {                                                         \
            _Pragma("unroll") for (int k = 0; k < 256 / NCH; k++) {                                  \
                _Pragma("unroll") for (int u = 0; u < NCH; u++) asm volatile(ASM : "+v"(acc[u]) : "v"(x), "v"(y)); \
            }                                                                                        \
        }                                                                                            \
        uint32_t s = 0;                                                                              \
        for (int u = 0; u < NCH; u++) s ^= acc[u];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                              \
    }
KERNEL(k_align1, 1, "v_alignbit_b32 %0, %0, %0, 7")
KERNEL(k_align2, 2, "v_alignbit_b32 %0, %0, %0, 7")
KERNEL(k_align4, 4, "v_alignbit_b32 %0, %0, %0, 7")
KERNEL(k_add1, 1, "v_add_u32 %0, %0, %1")
KERNEL(k_add2, 2, "v_add_u32 %0, %0, %1")
KERNEL(k_add4, 4, "v_add_u32 %0, %0, %1")
KERNEL(k_add3_1, 1, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_add3_2, 2, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_bitop1, 1, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_bitop2, 2, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_bitop4, 4, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")

typedef void (*kern_t)(uint32_t *, uint32_t, int);

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, kern_t k, int wps) {
        const int blocks = cus * wps, iters = 4000;
        k<<<blocks, 256>>>(out, 3, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, 256>>>(out, 3, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // one wave's instruction count: iters * 256; its time: ms
        printf("%-26s waves/SIMD %d: %6.2f ns per instruction of ONE wave  (= %.2f cycles at 2.1 GHz)\n", name, wps,
               ms * 1e6 / ((double)iters * 256), ms * 1e6 / ((double)iters * 256) * 2.1);
    };
    for (int wps : {1, 2}) {
        run("v_alignbit  chains=1", k_align1, wps);
        run("v_alignbit  chains=2", k_align2, wps);
        run("v_alignbit  chains=4", k_align4, wps);
        run("v_add_u32   chains=1", k_add1, wps);
        run("v_add_u32   chains=2", k_add2, wps);
        run("v_add_u32   chains=4", k_add4, wps);
        run("v_add3_u32  chains=1", k_add3_1, wps);
        run("v_add3_u32  chains=2", k_add3_2, wps);
        run("v_bitop3    chains=1", k_bitop1, wps);
        run("v_bitop3    chains=2", k_bitop2, wps);
        run("v_bitop3    chains=4", k_bitop4, wps);
    }
    return 0;
}
