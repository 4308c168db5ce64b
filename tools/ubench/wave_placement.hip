// wave_placement.hip -- where do the waves of small workgroups land?  (diagnostic; GPU box)
// The three-wave ladder pipeline (csrc/g1_pipe.hpp) wants every wave of a workgroup on a SIMD of its own.  Each wave of
// a grid of `groups` workgroups of `waves` waves records its HW_ID / XCC_ID, then spins ~0.3 ms so that the whole grid
// is resident at once.  Printed: workgroups whose waves share a SIMD, and the histogram of waves per (XCC, SE, CU, SIMD).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/wave_placement.hip -o tools/ubench/wave_placement && tools/ubench/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void k_mark(uint32_t *out, int lds_words) {
    extern __shared__ uint32_t sh[];
    if (threadIdx.x == 0 && lds_words) sh[0] = 1;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        uint32_t *rec = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 2;
        rec[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        rec[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 30000) {   // 100 MHz ticks: 0.3 ms
    }
}

static void run(int groups, int waves, int lds_bytes) {
    uint32_t *d;
    const size_t n = (size_t)groups * waves * 2;
    hipMalloc(&d, n * 4);
    hipFuncSetAttribute((const void *)k_mark, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k_mark, dim3(groups), dim3(64 * waves), lds_bytes, 0, d, lds_bytes / 4);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    hipFree(d);
    int shared_simd = 0;
    std::map<uint32_t, int> per_simd, per_cu;
    for (int g = 0; g < groups; g++) {
        int seen[4] = {0, 0, 0, 0};
        for (int w = 0; w < waves; w++) {
            const uint32_t id = h[((size_t)g * waves + w) * 2], xcc = h[((size_t)g * waves + w) * 2 + 1] & 15;
            const uint32_t simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            seen[simd]++;
            const uint32_t cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_simd[(cu_key << 2) | simd]++;
            per_cu[cu_key]++;
        }
        for (int s = 0; s < 4; s++)
            if (seen[s] > 1) {
                shared_simd++;
                break;
            }
    }
    int hist[8] = {0}, cuh[16] = {0};
    for (auto &kv : per_simd) hist[kv.second < 7 ? kv.second : 7]++;
    for (auto &kv : per_cu) cuh[kv.second < 15 ? kv.second : 15]++;
    printf("groups %d x %d waves, %d KB LDS: workgroups with two waves on one SIMD: %d; compute units used %zu; SIMDs holding 1/2/3/4+ waves: %d/%d/%d/%d; "
           "units holding [waves: count]:",
           groups, waves, lds_bytes / 1024, shared_simd, per_cu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
    for (int k = 1; k < 16; k++)
        if (cuh[k]) printf(" [%d: %d]", k, cuh[k]);
    printf("\n");
}

int main() {
    for (int lds : {0, 61 * 1024, 81 * 1024}) {
        run(160, 3, lds);
        run(168, 3, lds);
        run(336, 3, lds);
        run(336, 2, lds);
        run(672, 2, lds);
        run(336, 4, lds);
    }
    return 0;
}
