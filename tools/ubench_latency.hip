// How much ILP / TLP does the fp64 modular butterfly need on gfx950?
// Runs the mulmod chain with ILP independent chains per lane and W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr double P = 1695446975119361.0, PINV = 1.0 / P;
__device__ __forceinline__ double mulmod(double a, double w) {
  double h = a * w; double l = __builtin_fma(a, w, -h); double q = __builtin_rint(h * PINV);
  double r = __builtin_fma(-q, P, h); return r + l;
}
template <int ILP, int MODE>
__global__ __launch_bounds__(256) void k(double* out, double seed, int iters) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x * 7 + i * 12345;
  const double W = 123456789012345.0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (MODE == 0) x[i] = mulmod(x[i], W);
      else if (MODE == 1) x[i] = __builtin_fma(x[i], 0.999999, 1.0);         // dependent fma chain
      else { double t = mulmod(x[i], W); double a = x[(i + 1) % ILP]; x[i] = a + t; x[(i + 1) % ILP] = a - t; }
    }
  }
  double acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int ILP, int MODE>
int run(const char* name, int waves_per_simd, double* dout) {
  const int iters = 2048;
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(256), 0, 0, dout, 3.0, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(256), 0, 0, dout, 3.0 + r, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  double ops_per_wave = (double)iters * ILP;           // per-lane ops = wave-ops
  double cyc = ms * 1e-3 * 2.4e9 / ops_per_wave / waves_per_simd;  // SIMD cycles per wave-op
  printf("%-10s ILP=%d waves/SIMD=%d : %7.3f ms  %6.1f cyc per op per SIMD\n", name, ILP, waves_per_simd, ms, cyc);
  return 0;
}
int main() {
  double* d; CK(hipMalloc(&d, sizeof(double) * 256 * 256 * 8));
  for (int w : {1, 2, 3, 4, 8}) {
    run<1, 1>("fma_chain", w, d); run<4, 1>("fma_chain", w, d); run<8, 1>("fma_chain", w, d);
    run<1, 0>("mulmod", w, d); run<2, 0>("mulmod", w, d); run<4, 0>("mulmod", w, d); run<8, 0>("mulmod", w, d); run<16, 0>("mulmod", w, d);
    run<4, 2>("bfly", w, d); run<8, 2>("bfly", w, d);
  }
  return 0;
}
