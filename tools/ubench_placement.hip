// Where do the waves of 3-wave (or, 4th argument, 12-wave) workgroups land?  Records HW_ID (SIMD, CU, SE) and XCC_ID of
// every wave of a launch shaped like k_pbs (192 threads, `lds` bytes of dynamic LDS, a
// register budget of `REGS` VGPRs) while all workgroups are co-resident.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_placement ubench_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int REGS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_probe(uint32_t *out, long long spin)
{
    extern __shared__ unsigned char smem[];
    // burn registers so that the allocation matches k_pbs
    float r[REGS > 24 ? REGS - 24 : 1];
#pragma unroll
    for (int i = 0; i < (REGS > 24 ? REGS - 24 : 1); i++) r[i] = threadIdx.x * 0.5f + i;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < (REGS > 24 ? REGS - 24 : 1); i++) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < (REGS > 24 ? REGS - 24 : 1); i++) s += r[i];
    smem[threadIdx.x] = (unsigned char)s;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * WAVES + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc + (smem[threadIdx.x] == 255 ? 0x80000000u : 0u) * 0;
    }
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 1024;
    const int lds = argc > 2 ? atoi(argv[2]) : 35344;
    const int regs = argc > 3 ? atoi(argv[3]) : 168;
    const int W = argc > 4 && atoi(argv[4]) == 12 ? 12 : 3;
    uint32_t *d;
    hipMalloc(&d, sizeof(uint32_t) * 2 * W * wgs);
    auto kern = W == 12 ? k_probe<164, 12> : regs > 200 ? k_probe<250, 3> : regs > 130 ? k_probe<164, 3> : k_probe<100, 3>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncAttributes fa{};
    hipFuncGetAttributes(&fa, (const void *)kern);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, 64 * W, lds);
    printf("probe: regs %d, lds %d, occupancy %d WG/CU, %d workgroups\n", fa.numRegs, lds, nb, wgs);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * W), lds, 0, d, 100000000LL / 10 /* 100 MHz clock: 0.1 s */);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(2 * W * wgs);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per CU: waves per SIMD
    std::map<uint32_t, std::vector<int>> cu;
    std::map<std::vector<int>, int> wg_pattern;
    for (int g = 0; g < wgs; g++) {
        std::vector<int> pat;
        for (int w = 0; w < W; w++) {
            const uint32_t hw = h[2 * (W * g + w)], xcc = h[2 * (W * g + w) + 1] & 0xF;
            const int simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const uint32_t key = (xcc << 16) | (se << 8) | (sh << 4) | cuid;
            auto &v = cu[key];
            if (v.empty()) v.assign(4, 0);
            v[simd]++;
            pat.push_back(simd);
        }
        wg_pattern[pat]++;
    }
    std::map<std::vector<int>, int> hist;
    for (auto &kv : cu) hist[kv.second]++;
    printf("distinct CUs used: %zu\n", cu.size());
    printf("waves per SIMD (s0,s1,s2,s3) : number of CUs\n");
    for (auto &kv : hist) printf("  (%d,%d,%d,%d) : %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    printf("SIMDs of a workgroup's waves (w0,w1,...) : number of workgroups\n");
    for (auto &kv : wg_pattern) {
        printf("  (");
        for (size_t q = 0; q < kv.first.size(); q++) printf("%s%d", q ? "," : "", kv.first[q]);
        printf(") : %d\n", kv.second);
    }
    return 0;
}
