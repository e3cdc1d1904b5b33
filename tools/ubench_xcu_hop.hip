// What does it cost to hand a value from one CU to another on MI355X?  (could ONE bootstrap chain be split
// over two CUs?  A CMUX step of k_pbs_wide is 4 us; the two halves would have to exchange their partial sums once per step.)
// Two workgroups on different CUs play ping-pong through one 32-bit flag in global memory (release store / acquire spin load,
// agent scope): A writes 2 i - 1 and waits for 2 i, B waits for 2 i - 1 and writes 2 i.  One-way hop = time / (2 rounds).
// Workgroups are dealt round the eight XCDs in launch order, so blocks 0 and 1 sit on DIFFERENT XCDs (their L2s are not
// coherent with each other: the line travels through the fabric) and blocks 0 and 8 on the SAME XCD (one L2).  Each block asks
// for 100 KB of LDS so that no two share a CU.  A second pass moves a 4 KB payload (512 doubles: one polynomial half) with
// every hop - written before the flag, read after it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_xcu_hop tools/ubench_xcu_hop.hip && /tmp/ubench_xcu_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_pingpong(unsigned *flag, double *payload, int partner, int rounds, int with_payload,
                                                 unsigned long long *ticks, unsigned *xcc)
{
    extern __shared__ double lds[];
    const int me = blockIdx.x == 0 ? 0 : blockIdx.x == (unsigned)partner ? 1 : -1;
    if (me < 0) return;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[me] = id & 0xf;
    }
    double acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= rounds; i++) {
        const unsigned mine = me == 0 ? 2u * i - 1 : 2u * i, theirs = me == 0 ? 2u * i : 2u * i - 1;
        if (me == 1) {
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != theirs) {}
            if (with_payload)
                for (int q = threadIdx.x; q < 512; q += 64) acc += __hip_atomic_load(&payload[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (with_payload)
            for (int q = threadIdx.x; q < 512; q += 64) __hip_atomic_store(&payload[512 * (1 - me) + q], acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) __hip_atomic_store(flag, mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (me == 0) {
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != theirs) {}
            if (with_payload)
                for (int q = threadIdx.x; q < 512; q += 64) acc += __hip_atomic_load(&payload[512 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[me] = t1 - t0;
    lds[threadIdx.x] = acc;
}

int main()
{
    unsigned *flag, *xcc;
    double *payload;
    unsigned long long *ticks;
    CK(hipMalloc(&flag, 4));
    CK(hipMalloc(&xcc, 8));
    CK(hipMalloc(&payload, 1024 * sizeof(double)));
    CK(hipMalloc(&ticks, 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    const int rounds = 20000;
    for (int with_payload = 0; with_payload < 2; with_payload++)
        for (int partner : {1, 8, 9}) {
            CK(hipMemset(flag, 0, 4));
            CK(hipMemset(payload, 0, 1024 * sizeof(double)));
            hipLaunchKernelGGL(k_pingpong, dim3(partner + 1), dim3(64), 100 * 1024, 0, flag, payload, partner, rounds, with_payload, ticks, xcc);
            CK(hipDeviceSynchronize());
            unsigned long long t[2];
            unsigned x[2];
            CK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
            printf("{\"blocks\": [0, %d], \"xcc\": [%u, %u], \"payload_bytes_per_hop\": %d, \"rounds\": %d, \"one_way_hop_ns\": %.1f}\n", partner, x[0], x[1],
                   with_payload ? 4096 : 0, rounds, (double)t[0] * 10.0 / (2.0 * rounds));
        }
    return 0;
}
