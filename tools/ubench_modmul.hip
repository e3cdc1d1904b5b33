// Micro-benchmark: which arithmetic should the NTT use on gfx950?
// Measures sustained throughput (G lane-ops/s, and wave-instr cycles per SIMD)
// of the integer / fp64 primitives and of three complete modular multipliers.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_modmul.hip -o tools/ubench_modmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int ITER = 4096;
constexpr int ILP = 4;

// ---------- Goldilocks p = 2^64 - 2^32 + 1 ----------
__device__ __forceinline__ u64 gl_reduce128(u64 lo, u64 hi) {
  // hi = hh*2^32 + hl ; 2^64 = 2^32-1, 2^96 = -1 (mod p)
  u32 hh = (u32)(hi >> 32), hl = (u32)hi;
  u64 t = lo - hh;
  if (lo < hh) t -= 0xFFFFFFFFull;           // borrow: subtract epsilon
  u64 m = (u64)hl * 0xFFFFFFFFull;            // hl*(2^32-1)
  u64 r = t + m;
  if (r < m) r += 0xFFFFFFFFull;              // carry: add epsilon
  return r;                                    // in [0, 2^64), congruent
}
__device__ __forceinline__ u64 gl_mul(u64 a, u64 b) {
  u64 lo = a * b;
  u64 hi = __umul64hi(a, b);
  return gl_reduce128(lo, hi);
}
__device__ __forceinline__ u64 gl_add(u64 a, u64 b) {
  u64 r = a + b;
  if (r < a) r += 0xFFFFFFFFull;
  return r;
}
__device__ __forceinline__ u64 gl_sub(u64 a, u64 b) {
  u64 r = a - b;
  if (a < b) r -= 0xFFFFFFFFull;
  return r;
}

// ---------- fp64 modmul, p < 2^50, centred operands ----------
struct FpMod { double p, pinv; };
__device__ __forceinline__ double fp_mulmod(double a, double w, double p, double pinv) {
  double h = a * w;
  double l = __fma_rn(a, w, -h);
  double q = __builtin_rint(h * pinv);
  double r = __fma_rn(-q, p, h);
  return r + l;
}
// variant with precomputed w/p (Shoup-like): q from a*(w*pinv)
__device__ __forceinline__ double fp_mulmod_shoup(double a, double w, double wp, double p) {
  double h = a * w;
  double l = __fma_rn(a, w, -h);
  double q = __builtin_rint(a * wp);
  double r = __fma_rn(-q, p, h);
  return r + l;
}

// ---------- 31-bit prime, Shoup mul (w' = floor(w*2^32/p)) ----------
__device__ __forceinline__ u32 shoup_mul(u32 a, u32 w, u32 wp, u32 p) {
  u32 q = __umulhi(a, wp);
  u32 r = a * w - q * p;
  return r >= p ? r - p : r;
}
// ---------- 31-bit Montgomery ----------
__device__ __forceinline__ u32 mont_mul(u32 a, u32 b, u32 p, u32 pinv) {
  u64 t = (u64)a * b;
  u32 m = (u32)t * pinv;
  u32 u = __umulhi(m, p);
  u32 hi = (u32)(t >> 32);
  u32 r = hi - u;
  return hi < u ? r + p : r;
}

enum Op { MUL_LO, MUL_HI, MAD64, MUL64, MULHI64, FMA64, ADD64F, MULF64, RNDNE, CVT_I2D, CVT_D2I,
          GL_MUL, GL_ADD, GL_BFLY, FP_MUL, FP_MUL_SHOUP, FP_BFLY, SHOUP, MONT, FMA32, NOPS };

template <int OP>
__global__ __launch_bounds__(256) void bench(u64* out, u64 seed) {
  u64 tid = blockIdx.x * 256 + threadIdx.x;
  u64 x[ILP]; double d[ILP]; u32 s[ILP]; float f[ILP];
#pragma unroll
  for (int i = 0; i < ILP; i++) {
    x[i] = seed * (tid + 1) + i * 0x9E3779B97F4A7C15ull;
    d[i] = (double)((x[i] >> 15) & ((1ull << 49) - 1));
    s[i] = (u32)x[i] | 1;
    f[i] = (float)(s[i] & 0xffff);
  }
  const u64 wq = seed ^ 0xD1B54A32D192ED03ull;
  const double P = 1125899906826241.0;        // 2^50 - 2^14*... (any odd ~2^50 value; primality irrelevant for timing)
  const double PINV = 1.0 / P;
  const double W = 123456789012345.0, WP = W / P;
  const u32 p32 = 2013265921u, w32 = (u32)wq % p32, wp32 = (u32)(((u64)w32 << 32) / p32), pinv32 = 2013265919u;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == MUL_LO) s[i] = s[i] * s[(i + 1) % ILP];
      else if (OP == MUL_HI) s[i] = __umulhi(s[i], s[(i + 1) % ILP]) | 0x80000001u;
      else if (OP == MAD64) x[i] = (u64)(u32)x[i] * (u32)x[(i + 1) % ILP] + x[i];
      else if (OP == MUL64) x[i] = x[i] * (x[(i + 1) % ILP] | 1);
      else if (OP == MULHI64) x[i] = __umul64hi(x[i], x[(i + 1) % ILP]) | 0x8000000000000001ull;
      else if (OP == FMA64) d[i] = __fma_rn(d[i], 0.999999, d[(i + 1) % ILP]);
      else if (OP == ADD64F) d[i] = d[i] + d[(i + 1) % ILP];
      else if (OP == MULF64) d[i] = d[i] * 1.0000001;
      else if (OP == RNDNE) d[i] = __builtin_rint(d[i]) * 0.5;   // rint + mul
      else if (OP == CVT_I2D) d[i] += (double)(int)s[i];          // cvt + add
      else if (OP == CVT_D2I) s[i] += (u32)(int)d[i];             // cvt + iadd
      else if (OP == GL_MUL) x[i] = gl_mul(x[i], wq);
      else if (OP == GL_ADD) x[i] = gl_add(x[i], x[(i + 1) % ILP]);
      else if (OP == GL_BFLY) { u64 t = gl_mul(x[i], wq); u64 a = x[(i + 1) % ILP]; x[i] = gl_add(a, t); x[(i + 1) % ILP] = gl_sub(a, t); }
      else if (OP == FP_MUL) d[i] = fp_mulmod(d[i], W, P, PINV);
      else if (OP == FP_MUL_SHOUP) d[i] = fp_mulmod_shoup(d[i], W, WP, P);
      else if (OP == FP_BFLY) { double t = fp_mulmod_shoup(d[i], W, WP, P); double a = d[(i + 1) % ILP]; d[i] = a + t; d[(i + 1) % ILP] = a - t; }
      else if (OP == SHOUP) s[i] = shoup_mul(s[i], w32, wp32, p32);
      else if (OP == MONT) s[i] = mont_mul(s[i], w32, p32, pinv32);
      else if (OP == FMA32) f[i] = __fmaf_rn(f[i], 0.99999f, f[(i + 1) % ILP]);
    }
  }
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc += x[i] + (u64)d[i] + s[i] + (u64)f[i];
  out[tid] = acc;
}

template <int OP>
int run(const char* name, int ops_per_iter, u64* dout) {
  const int blocks = 256 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, dout, 0x1234567ull);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 3;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, dout, 0x1234567ull + r);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  double laneops = (double)blocks * 256 * ITER * ILP * ops_per_iter;
  double gops = laneops / (ms * 1e-3) / 1e9;
  // cycles per wave-op per SIMD at 2.4 GHz: 1024 SIMDs
  double waveops_per_simd = laneops / 64.0 / 1024.0;
  double cyc = (ms * 1e-3 * 2.4e9) / waveops_per_simd;
  printf("%-14s %8.3f ms  %10.1f Gop/s  %6.2f cyc/wave-op/SIMD (@2.4GHz)\n", name, ms, gops, cyc);
  return 0;
}

int main() {
  u64* dout; CK(hipMalloc(&dout, sizeof(u64) * 256 * 8 * 256));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  run<FMA32>("fma_f32", 1, dout);
  run<MUL_LO>("mul_lo_u32", 1, dout);
  run<MUL_HI>("mul_hi_u32", 1, dout);
  run<MAD64>("mad_u64_u32", 1, dout);
  run<MUL64>("mul_u64", 1, dout);
  run<MULHI64>("mulhi_u64", 1, dout);
  run<FMA64>("fma_f64", 1, dout);
  run<ADD64F>("add_f64", 1, dout);
  run<MULF64>("mul_f64", 1, dout);
  run<RNDNE>("rint+mul_f64", 1, dout);
  run<CVT_I2D>("cvt_i2d+add", 1, dout);
  run<CVT_D2I>("cvt_d2i+iadd", 1, dout);
  run<GL_MUL>("gl_mul", 1, dout);
  run<GL_ADD>("gl_add", 1, dout);
  run<GL_BFLY>("gl_bfly", 1, dout);
  run<FP_MUL>("fp_mul", 1, dout);
  run<FP_MUL_SHOUP>("fp_mul_shoup", 1, dout);
  run<FP_BFLY>("fp_bfly", 1, dout);
  run<SHOUP>("shoup31", 1, dout);
  run<MONT>("mont31", 1, dout);
  return 0;
}
