// ntt_fp64.h — exact negacyclic NTT over a 51-bit prime field, computed with the
// fp64 FMA pipe of CDNA4.
//
// Why fp64: measured on MI355X (profiles/r01_ubench_modmul.txt) a complete modular
// butterfly costs ~34 SIMD-cycles per wave with fp64 FMA error-free products, ~137
// with a 64-bit Goldilocks multiply and ~64 with two 31-bit Shoup primes: 32x32->64
// integer multiplies are quarter-rate on gfx950 while v_fma_f64 is half-rate.
//
// Every value is an INTEGER held exactly in a double (|v| < 2^53).  Products are
// formed error-free (h = a*w rounded, l = fma(a,w,-h) its exact error) and reduced
// with a rounded quotient, so results are exact residues: the transform is an exact
// NTT, and the polynomial products it yields are bit-identical to schoolbook
// arithmetic mod 2^32 as long as the true integer result fits in (-p/2, p/2).
//
// One wave transforms one polynomial: lane holds E = N/64 coefficients, the log2(N)
// radix-2 stages are fused into three in-register blocks separated by two
// wave-private LDS transposes (no workgroup barrier inside a transform).  Several
// polynomials of one wave are transformed together so that their LDS round trips
// and fp64 dependency chains overlap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace helm {

// Fields: primes p = b^4 + 1 (round 4).  b is then a primitive eighth root of unity, so the roots of unity of the first
// two transform stages are the short integers b^2, b, b^3, and stages on decomposition digits lose their modular
// reductions.  Rounds 1-3 used arbitrary primes of the same sizes (0x6060002B00001, 2^50.6, recentred at every block
// boundary: 2^53 / p = 5.3; 0x2424DD2F20001 and 0x24007A8500001, 2^49.2, lazy: 2^53 / p = 14.2, forward outputs <= 7.1 p,
// hand-over sums <= 11.4 p); the bounds of the present fields are recomputed in tests/test_lazy_bounds.py.
//   FpG   5072^4 + 1 = 2^49.23  lazy    boolean kernels (N = 512 sets whose exact products fit) and CRT prime 0 of the 64-bit ones
//   FpG2  5096^4 + 1 = 2^49.26  lazy    CRT prime 1 of the 64-bit kernels
//   FpH   6432^4 + 1 = 2^50.60  strict  boolean kernels at N = 1024 and for larger N = 512 sets
//   FpI   5440^4 + 1 = 2^49.64  lazy    boolean kernels at N = 1024 when the loaded key's own bound fits (round 5)

// The lazy field of the BOOLEAN kernels (round 4): p = b^4 + 1 with b = 5072 = 2^12.3 (generator 3; 2^16 | p - 1).
// b is a primitive EIGHTH root of unity, so the twiddles of a transform's first two stages - psi^(N/2) and psi^(N/4),
// psi^(3N/4), with psi chosen such that psi^(N/4) = b (helm_hip_ctx_create) - are b^2, b, b^3: 25, 12 and 37 bits.
// On decomposition digits (|d| <= 2^(logB-1)) every product of those two stages is an exact double far inside (-p/2, p/2):
// the radix-4 butterfly on four digits is 10 plain operations instead of 6 modular multiplications and 8 additions
// (fwd_top2_digits: 20 instead of 56 per transform and polynomial at N = 512).  p / 2 = 2^48.23 still covers tfhe boolean
// DEFAULT's exact products (2^48.17); 2^53 / p = 13.6 against forward outputs <= 4.9 p, hand-over sums <= 9.4 p, inverse
// sums <= 8 p (tests/test_lazy_bounds.py).
struct FpG {
    static constexpr double P = 661785091833857.0;
    static constexpr uint64_t P_U64 = 661785091833857ull;
    static constexpr uint64_t GEN = 3;
    static constexpr bool LAZY = true;
    static constexpr double B1 = 5072.0, B2 = 25725184.0, B3 = 130478133248.0; // b, b^2, b^3 (b^4 = -1)
};
// FpG's partner in the CRT pair of the 64-bit-torus kernels (helm_shortint.hip): p = 5096^4 + 1 = 2^49.26 (generator 3,
// 2^12 | p - 1: N <= 2048).  There the digits are up to 2^22, so only the FOURTH root b^2 (25 bits) is short enough:
// digit x psi^(N/2) is an exact double inside (-p/2, p/2) - stage 1 of a forward transform as one multiplication.
struct FpG2 {
    static constexpr double P = 674400179654657.0;
    static constexpr uint64_t P_U64 = 674400179654657ull;
    static constexpr uint64_t GEN = 3;
    static constexpr bool LAZY = true;
    static constexpr double B1 = 5096.0, B2 = 25969216.0, B3 = 132339124736.0;
};
// The same for the 51-bit field of the boolean kernels (N = 1024 sets, and N = 512 sets too large for FpG):
// p = 6432^4 + 1 = 2^50.6 (generator 5; 2^12 | p - 1), 1 % above Fp<51>'s prime: p/2 covers the same sets, 2^53 / p = 5.26
// (Fp<51>: 5.32) with every bound of the recentring build at most 4.5 p.  b, b^2, b^3 = 13, 26, 38 bits.
struct FpH {
    static constexpr double P = 1711528530149377.0;
    static constexpr uint64_t P_U64 = 1711528530149377ull;
    static constexpr uint64_t GEN = 5;
    static constexpr bool LAZY = false;
    static constexpr double B1 = 6432.0, B2 = 41370624.0, B3 = 266095853568.0;
};
// Round 5: a LAZY field for the N = 1024 sets (the reference's CUDA parameters, helm.rs:141-146): p = 5440^4 + 1 = 2^49.64
// (generator 3; 2^24 | p - 1), 2^53 / p = 10.28.  Its half, 2^48.64, is BELOW the worst case of helm_cuda's exact products
// ((k+1) l N B/2 2^31 = 2^49.58, every key coefficient at 2^31 with aligned signs) and ABOVE what a loaded key can produce:
// |sum| <= B/2 x the largest l1-norm of a key column (about N (k+1) l 2^30 = 2^48.58 for a key of uniform masks), which
// helm_hip_load_bootstrap_key computes for the key at hand - an exact guarantee for that key and every input; a key that
// does not fit keeps FpH.  Bounds (tests/test_lazy_bounds.py): forward outputs <= 6.9 p, hand-over sums of the six products
// <= 9.1 p, inverse sums <= 8 p, all below 10.28 p: no recentring in the forward transforms, the products, their sums or the
// four-stage inverse block - 112 of FpH's 160 recentrings per wave-step (of 3,785 vector instructions) are gone.
struct FpI {
    static constexpr double P = 875781160960001.0;
    static constexpr uint64_t P_U64 = 875781160960001ull;
    static constexpr uint64_t GEN = 3;
    static constexpr bool LAZY = true;
    static constexpr double B1 = 5440.0, B2 = 29593600.0, B3 = 160989184000.0;
};
// Round 6: a 46-bit CRT pair for 64-bit-torus sets with short exact products - the set the reference binary installs for LUT
// mode, PARAM_MESSAGE_1_CARRY_1_KS_PBS (src/bin/helm.rs:301: k = 3, N = 512, one level of 18 bits): its exact products are
// below B/2 x the largest l1-norm of a column of the LOADED key, about 2^90.0 for a generated key (worst case of the set:
// 2^91), and p p' / 2 = 2^90.62 covers that - helm_si_load_bootstrap_key computes the bound for the key at hand, a key that
// does not fit keeps the 49-bit pair (exact either way: identical ciphertexts).  What the smaller primes buy is HEADROOM:
// 2^53 / p = 160 and 132 instead of 13.6.  b^3 is 34.3 / 34.5 bits, so on 17-bit digits BOTH leading stages of a forward
// transform are the plain radix-4 butterfly (fwd_top2_digits: every term digit x root <= 2^51.5, an exact double; the 49-bit
// pair can only do stage 1), nothing in a forward transform, the products or their sums needs a recentring, and the inverse
// transform recentres ONE slot at each of its two transposes instead of eight (ntt_inverse, WIDE).
//   FpJ   2736^4 + 1 = 2^45.67 (generator 5; 2^16 | p - 1)      FpJ2  2872^4 + 1 = 2^45.95 (generator 3; 2^12 | p - 1)
struct FpJ {
    static constexpr double P = 56035644604417.0;
    static constexpr uint64_t P_U64 = 56035644604417ull;
    static constexpr uint64_t GEN = 5;
    static constexpr bool LAZY = true;
    static constexpr double B1 = 2736.0, B2 = 7485696.0, B3 = 20480864256.0;
};
struct FpJ2 {
    static constexpr double P = 68035838611457.0;
    static constexpr uint64_t P_U64 = 68035838611457ull;
    static constexpr uint64_t GEN = 3;
    static constexpr bool LAZY = true;
    static constexpr double B1 = 2872.0, B2 = 8248384.0, B3 = 23689358848.0;
};
static_assert(2736ull * 2736 * 2736 * 2736 + 1 == FpJ::P_U64 && 2872ull * 2872 * 2872 * 2872 + 1 == FpJ2::P_U64, "p = b^4 + 1");
static_assert(2736ull * 2736 == 7485696ull && 7485696ull * 2736 == 20480864256ull && 2872ull * 2872 == 8248384ull &&
                  8248384ull * 2872 == 23689358848ull, "b^2, b^3");
// fields with 2^53 / p >= 128: values may stay unreduced through whole transforms (see FpJ)
template <typename F> struct wide_headroom : std::false_type {};
template <> struct wide_headroom<FpJ> : std::true_type {};
template <> struct wide_headroom<FpJ2> : std::true_type {};
// ntt_inverse's LEAN form (half of the recentrings at its two transposes) rests on slot-class bounds derived with
// 2^53 / p >= 13.6 (tests/test_lazy_bounds.py): the 49.2-bit fields only.  FpI (2^53 / p = 10.28) recentres every slot.
template <typename F> struct lean_inverse_ok : std::false_type {};
template <typename F> struct has_short_roots : std::false_type {};
template <> struct has_short_roots<FpI> : std::true_type {};
template <> struct has_short_roots<FpG> : std::true_type {};
template <> struct has_short_roots<FpG2> : std::true_type {};
template <> struct has_short_roots<FpH> : std::true_type {};
template <> struct has_short_roots<FpJ> : std::true_type {};
template <> struct has_short_roots<FpJ2> : std::true_type {};
template <> struct lean_inverse_ok<FpG> : std::true_type {};
template <> struct lean_inverse_ok<FpG2> : std::true_type {};

// -DHELM_CHECK_BOUNDS: the contracts the lazy arithmetic rests on, checked at run time (a debug build, one translation unit:
// `make libhelm_hip_check.so`, tests/test_gpu_bounds_check.py).  Every value is an exact integer held in a double, which is
// only true below 2^53; the lean inverse transform additionally wants its inputs recentred (|x| <= p/2).  A violation is
// COUNTED (helm_hip_bound_violations), never trapped: a trap would take the GPU context down.
//   slot 0  mulmod: |a| >= 2^53        1  reduce: |a| >= 2^53        2  a butterfly sum or difference >= 2^53
//   slot 3  ntt_inverse (lean form) entered with |x| > p/2           4  a lifted value outside to_torus32's range (2^51)
#ifdef HELM_CHECK_BOUNDS
static __device__ unsigned int g_helm_bound_violations[8];
#define HELM_BOUND(cond, slot)                                                           \
    do {                                                                                 \
        if (!(cond)) atomicAdd(&g_helm_bound_violations[slot], 1u);                      \
    } while (0)
#else
#define HELM_BOUND(cond, slot) ((void)0)
#endif

// a*w mod p for integers |a| < 2^53, |w| <= p/2.  Result r == a*w (mod p) exactly,
// |r| <= (0.5 + 0.75 * |a| * 2^-52) * p  (<= 2p for any admissible a).
template <typename F>
__device__ __forceinline__ double mulmod(double a, double w)
{
    constexpr double PINV = 1.0 / F::P;
    HELM_BOUND(__builtin_fabs(a) < 0x1p53, 0);
    double h = a * w;
    double l = __builtin_fma(a, w, -h);
    double q = __builtin_rint(h * PINV);
    double r = __builtin_fma(-q, F::P, h);
    return r + l;
}

// a mod p, centred: |result| <= (0.5 + eps) * p.
template <typename F>
__device__ __forceinline__ double reduce(double a)
{
    constexpr double PINV = 1.0 / F::P;
    HELM_BOUND(__builtin_fabs(a) < 0x1p53, 1);
    double q = __builtin_rint(a * PINV);
    return __builtin_fma(-q, F::P, a);
}
// recentre only where the field has no headroom to skip it
template <typename F>
__device__ __forceinline__ double reduce_unless_lazy(double a)
{
    if constexpr (F::LAZY) return a;
    else return reduce<F>(a);
}

// Exact integer in a double (|v| < 2^51) -> v mod 2^32.
__device__ __forceinline__ uint32_t to_torus32(double v)
{
    HELM_BOUND(__builtin_fabs(v) < 0x1p51, 4);
    return (uint32_t)__double2loint(v + 6755399441055744.0 /* 1.5 * 2^52 */);
}

// Wave-private LDS hand-off: LDS operations of one wave execute in order, so no
// s_barrier is needed; the LDS-only fence just stops the compiler from reordering
// the accesses (global loads in flight are NOT affected, unlike __syncthreads()).
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

// Workgroup barrier ordering LDS only: global prefetches stay in flight across it.
__device__ __forceinline__ void lds_block_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Key words are read with buffer loads: one resource descriptor (scalar registers) for the whole
// bootstrapping key, a scalar byte offset per (step, row, column, level), one lane-offset
// register and an immediate per word - no per-load 64-bit vector address arithmetic.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct KeyBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane16; // lane * 16 bytes
    __device__ __forceinline__ void init(const void *base, size_t bytes, int lane)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                                 (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), 0x00020000);
        lane16 = lane * 16;
    }
    // double2 at byte offset soff (wave-uniform) + imm (compile-time) + lane * 16
    __device__ __forceinline__ double2 load(unsigned soff, int imm) const
    {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16 + imm, (int)soff, 0);
        double2 d;
        __builtin_memcpy(&d, &v, 16);
        return d;
    }
};

// Wave-private LDS accumulate (ds_add_f64, no return value).
__device__ __forceinline__ void lds_add(double *p, double v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// LDS accumulate shared by the waves of a workgroup (atomic ds_add_f64).
__device__ __forceinline__ void lds_add_wg(double *p, double v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// twiddles one lane needs for a block: sum over its stages of E >> (eb+1)
constexpr int ntw_count(int E, int shift, int sb_lo, int sb_hi)
{
    int c = 0;
    for (int sb = sb_lo; sb <= sb_hi; sb++) c += E >> (sb - shift + 1);
    return c;
}

template <int LOGN>
struct Geo {
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGE = LOGN - 6;
    static constexpr int E = 1 << LOGE;           // coefficients per lane
    static constexpr int BA = (LOGN == 9) ? 3 : 4; // stages in the first block
    static constexpr int BC = 3;                   // stages in the last block
    static constexpr int BB = LOGN - BA - BC;      // stages in the middle block
    static constexpr int XPAD = N + 64;            // padded exchange buffer (doubles)
    static_assert(LOGN == 9 || LOGN == 10 || LOGN == 11, "supported polynomial sizes");
    static_assert(BA <= LOGE && BB <= LOGE && BC <= LOGE, "block does not fit in registers");

    static constexpr int TWA = ntw_count(E, 6, LOGN - BA, LOGN - 1);
    static constexpr int TWB = ntw_count(E, BC, BC, BC + BB - 1);
    static constexpr int TWC = ntw_count(E, 0, 0, BC - 1);
    static constexpr int NTW = TWA + TWB + TWC; // per direction

    // Three register layouts: which coefficient index j lane holds in slot e.
    __device__ static __forceinline__ int jA(int lane, int e) { return (e << 6) | lane; }
    __device__ static __forceinline__ int jB(int lane, int e)
    {
        return ((lane >> BC) << (BC + LOGE)) | (e << BC) | (lane & ((1 << BC) - 1));
    }
    __device__ static __forceinline__ int jC(int lane, int e) { return (lane << LOGE) | e; }
    // LDS paddings making both sides of each transpose bank-conflict free.
    __device__ static __forceinline__ int pad1(int j) { return j + ((j >> (LOGN - 3)) << 3); }
    __device__ static __forceinline__ int pad2(int j) { return j + (j >> LOGE); }
    // The same padded indices as (per-lane base) + (compile-time offset of slot e), so that every
    // transpose access is one address register plus an immediate:
    //   pad1(jA) = baseA(lane) + offA1(e)    pad1(jB) = baseB(lane) + offB1(e)
    //   pad2(jB) = baseB(lane) + offB2(e)    pad2(jC) = baseC(lane) + e
    __device__ static __forceinline__ int baseA(int lane) { return lane; }
    __device__ static __forceinline__ int baseB(int lane)
    {
        return ((lane >> BC) << (LOGN - 3)) + ((lane >> BC) << 3) + (lane & ((1 << BC) - 1));
    }
    __device__ static __forceinline__ int baseC(int lane) { return lane * (E + 1); }
    static constexpr int offA1(int e) { return (e << 6) + ((e >> (LOGN - 9)) << 3); }
    static constexpr int offB1(int e) { return e << BC; }
    static constexpr int offB2(int e) { return (e << BC) + (e >> (LOGE - BC)); }
    static_assert(BC == 3, "the base/offset forms assume 8-lane groups in layout B");
};

// Twiddle sources.  Twiddle of the butterfly on stride bit sb for coefficient j is
// table[(N >> (sb+1)) + (j >> (sb+1))] (table = bit-reversed powers of psi).  A source is
// asked with get(sb, hi, cnt, idx, slot): hi = which of the stage's cnt twiddles of this
// lane, idx = table index, slot = running count in consumption order.
//   TwMem    fetches table[idx] (LDS or global table)
//   TwLane   block A twiddles are the same for every lane (scalar registers); those of
//            blocks B and C come from a lane-major LDS table [slot][64] of the FORWARD
//            twiddles: one base address register + immediate offsets.  The inverse
//            direction reads the same table mirrored: psi^-i = -psi^(N-i), which in the
//            bit-reversed table is inv[2^s + r] = -fwd[2^(s+1) - 1 - r], i.e. the twiddle
//            (cnt-1-hi) of lane 63-lane; inv_block absorbs the sign by swapping its
//            subtraction (MIRROR).
struct TwMem {
    static constexpr bool MIRROR = false;
    const double *t;
    __device__ __forceinline__ double get(int, int, int, int idx, int) const { return t[idx]; }
};
// forward-order slot of twiddle (sb, hi) among one lane's twiddles (block A, then B, then C)
template <int LOGN>
constexpr int tw_fwd_slot(int sb, int hi)
{
    using G = Geo<LOGN>;
    int slot = 0;
    for (int s = LOGN - 1; s >= LOGN - G::BA; s--) {
        if (s == sb) return slot + hi;
        slot += G::E >> (s - 6 + 1);
    }
    for (int s = G::BC + G::BB - 1; s >= G::BC; s--) {
        if (s == sb) return slot + hi;
        slot += G::E >> (s - G::BC + 1);
    }
    for (int s = G::BC - 1; s >= 0; s--) {
        if (s == sb) return slot + hi;
        slot += G::E >> (s + 1);
    }
    return -1;
}
// table index of lane-table slot s (0 .. TWB+TWC-1) for a lane
template <int LOGN>
__device__ inline int tw_lane_index(int s, int lane)
{
    using G = Geo<LOGN>;
    int slot = 0;
    for (int sb = G::BC + G::BB - 1; sb >= G::BC; sb--) {
        const int eb = sb - G::BC, cnt = G::E >> (eb + 1);
        if (s < slot + cnt) {
            const int jh = G::jB(lane, 0) | ((s - slot) << (eb + 1 + G::BC));
            return (G::N >> (sb + 1)) + (jh >> (sb + 1));
        }
        slot += cnt;
    }
    for (int sb = G::BC - 1; sb >= 0; sb--) {
        const int cnt = G::E >> (sb + 1);
        if (s < slot + cnt) {
            const int jh = G::jC(lane, 0) | ((s - slot) << (sb + 1));
            return (G::N >> (sb + 1)) + (jh >> (sb + 1));
        }
        slot += cnt;
    }
    return 0;
}
template <int LOGN, bool MIRROR_>
struct TwLane {
    static constexpr bool MIRROR = MIRROR_;
    using G = Geo<LOGN>;
    const double *base; // LDS table + lane (forward) or + 63 - lane (mirrored)
    double ua[G::TWA];  // block A: lane-uniform
    __device__ __forceinline__ double get(int sb, int hi, int cnt, int, int) const
    {
        const int fs = tw_fwd_slot<LOGN>(sb, MIRROR ? cnt - 1 - hi : hi);
        if (fs < G::TWA) return ua[fs];
        return base[(fs - G::TWA) * 64];
    }
    // fill the uniform part from the global forward table (compile-time indices: scalar loads)
    __device__ __forceinline__ void fill_uniform(const double *__restrict__ table)
    {
        int slot = 0;
#pragma unroll
        for (int sb = LOGN - 1; sb >= LOGN - G::BA; sb--) {
#pragma unroll
            for (int hi = 0; hi < (G::E >> (sb - 6 + 1)); hi++) ua[slot++] = table[(G::N >> (sb + 1)) + hi];
        }
    }
};

// TwLane with the lane's block-B and block-C twiddles of the FORWARD direction copied into registers once per kernel
// (they do not change from step to step): TWB + TWC fewer LDS reads per forward transform, for kernels that have the
// registers.  (The inverse direction keeps reading the table mirrored.)
template <int LOGN, bool MIRROR_ = false>
struct TwLaneReg {
    static constexpr bool MIRROR = MIRROR_;
    using G = Geo<LOGN>;
    double ua[G::TWA];          // block A: lane-uniform
    double c[G::TWB + G::TWC];  // blocks B and C: this lane's (MIRROR: lane 63 - lane's, read in mirrored slot order)
    __device__ __forceinline__ void load(const TwLane<LOGN, MIRROR_> &t)
    {
#pragma unroll
        for (int r = 0; r < G::TWA; r++) ua[r] = t.ua[r];
#pragma unroll
        for (int r = 0; r < G::TWB + G::TWC; r++) c[r] = t.base[r * 64];
    }
    __device__ __forceinline__ double get(int sb, int hi, int cnt, int, int) const
    {
        const int fs = tw_fwd_slot<LOGN>(sb, MIRROR ? cnt - 1 - hi : hi);
        if (fs < G::TWA) return ua[fs];
        return c[fs - G::TWA];
    }
};
template <int LOGN> using TwLaneFwdReg = TwLaneReg<LOGN, false>;

// Index table for blocks A and B (entries below N >> BC: few, read with few distinct
// addresses per instruction) + lane-major table for block C, whose per-lane indices stride
// by E through an index table (E-way bank conflicts for E = 16, 32).
template <int LOGN, bool MIRROR_>
struct TwHybrid {
    static constexpr bool MIRROR = MIRROR_;
    using G = Geo<LOGN>;
    const double *t;    // LDS index table, entries [0, N >> BC)
    const double *base; // LDS lane table of block C [TWC][64] + lane (forward) or + 63 - lane (mirrored)
    __device__ __forceinline__ double get(int sb, int hi, int cnt, int idx, int) const
    {
        if (sb < G::BC) {
            const int fs = tw_fwd_slot<LOGN>(sb, MIRROR ? cnt - 1 - hi : hi) - G::TWA - G::TWB;
            return base[fs * 64];
        }
        return t[MIRROR ? 3 * (G::N >> (sb + 1)) - 1 - idx : idx];
    }
};

// TwHybrid with the block-C twiddles of the lane held in registers (they do not change from step to step): TWC fewer LDS
// reads per transform, for kernels that have the registers.
template <int LOGN, bool MIRROR_>
struct TwHybridC {
    static constexpr bool MIRROR = MIRROR_;
    using G = Geo<LOGN>;
    const double *t; // LDS index table, entries [0, N >> BC)
    double c[G::TWC];
    __device__ __forceinline__ void load(const double *base) // base as TwHybrid's
    {
#pragma unroll
        for (int r = 0; r < G::TWC; r++) c[r] = base[r * 64];
    }
    __device__ __forceinline__ double get(int sb, int hi, int cnt, int idx, int) const
    {
        if (sb < G::BC) return c[tw_fwd_slot<LOGN>(sb, MIRROR ? cnt - 1 - hi : hi) - G::TWA - G::TWB];
        return t[MIRROR ? 3 * (G::N >> (sb + 1)) - 1 - idx : idx];
    }
};

// Fused radix-2 Cooley-Tukey stages on stride bits SB_HI..SB_LO (descending), all of
// which are register-slot bits (slot bit = stride bit - SHIFT), for M polynomials.
// PLAIN_TOP: the block starts with the transform's FIRST stage (stride bit LOGN-1, the one twiddle psi^(N/2) = +-b^2 of a
// b^4 + 1 field) on inputs short enough that the product is an exact double inside (-p/2, p/2): a plain multiplication.
template <typename F, int LOGN, int M, int SHIFT, int SB_HI, int SB_LO, int SLOT0, typename TW, bool PLAIN_TOP = false>
__device__ __forceinline__ void fwd_block(double (&x)[M][Geo<LOGN>::E], const TW &tw, int jbase)
{
    static_assert(!PLAIN_TOP || SB_HI == LOGN - 1, "the short twiddle is the first stage's");
    constexpr int E = Geo<LOGN>::E, N = Geo<LOGN>::N;
    int slot = SLOT0;
#pragma unroll
    for (int sb = SB_HI; sb >= SB_LO; sb--) {
        const int eb = sb - SHIFT;
#pragma unroll
        for (int hi = 0; hi < (E >> (eb + 1)); hi++) {
            const int jh = jbase | (hi << (eb + 1 + SHIFT));
            const double w = tw.get(sb, hi, E >> (eb + 1), (N >> (sb + 1)) + (jh >> (sb + 1)), slot++);
#pragma unroll
            for (int lo = 0; lo < (1 << eb); lo++) {
                const int e0 = (hi << (eb + 1)) | lo, e1 = e0 | (1 << eb);
#pragma unroll
                for (int m = 0; m < M; m++) {
                    double U = x[m][e0], V = (PLAIN_TOP && sb == SB_HI) ? x[m][e1] * w : mulmod<F>(x[m][e1], w);
                    x[m][e0] = U + V;
                    x[m][e1] = U - V;
                    HELM_BOUND(__builtin_fabs(x[m][e0]) < 0x1p53 && __builtin_fabs(x[m][e1]) < 0x1p53, 2);
                }
            }
        }
    }
}

// Gentleman-Sande stages on stride bits SB_LO..SB_HI (ascending).
template <typename F, int LOGN, int SHIFT, int SB_LO, int SB_HI, int SLOT0, typename TW>
__device__ __forceinline__ void inv_block(double (&x)[Geo<LOGN>::E], const TW &tw, int jbase)
{
    constexpr int E = Geo<LOGN>::E, N = Geo<LOGN>::N;
    int slot = SLOT0;
#pragma unroll
    for (int sb = SB_LO; sb <= SB_HI; sb++) {
        const int eb = sb - SHIFT;
        if (!F::LAZY && sb - SB_LO == 2 && SB_HI - SB_LO == 3) {
            // 4-stage block: the pure-sum path has doubled twice; recentre (51-bit field: 2^53 = 5.3 p.
            // The lazy 49-bit fields hold four doublings from 0.5 p: sums <= 8 p, differences fed to
            // the multiplications <= 8 p at the last stage, below 2^53 = 14.2 p).
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = reduce<F>(x[e]);
        }
#pragma unroll
        for (int hi = 0; hi < (E >> (eb + 1)); hi++) {
            const int jh = jbase | (hi << (eb + 1 + SHIFT));
            const double w = tw.get(sb, hi, E >> (eb + 1), (N >> (sb + 1)) + (jh >> (sb + 1)), slot++);
#pragma unroll
            for (int lo = 0; lo < (1 << eb); lo++) {
                const int e0 = (hi << (eb + 1)) | lo, e1 = e0 | (1 << eb);
                double U = x[e0], V = x[e1];
                x[e0] = U + V;
                HELM_BOUND(__builtin_fabs(x[e0]) < 0x1p53 && __builtin_fabs(U - V) < 0x1p53, 2);
                x[e1] = mulmod<F>(TW::MIRROR ? V - U : U - V, w);
            }
        }
    }
}

// Forward negacyclic NTT of M polynomials held by one wave.
// in : x[m][e] = coefficient jA(lane,e), |x| <= 0.5p.
// out: x[m][e] = transform word at position jC(lane,e) of the bit-reversed output,
//      |x| <= ~3p (Fp51) / ~7.1p (Fp49, lazy); not recentred: the pointwise product absorbs it.
// xbuf: wave-private LDS scratch of M * Geo::XPAD doubles.
// PRIO > 0: the wave enters at issue priority PRIO and steps down by one after each of the first two
// blocks (waves that share a SIMD and run the same phase then advance block by block, see k_pbs).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// before_last: called between the second transpose and the last block (both LDS round trips behind, a block of pure
// arithmetic ahead): the place to issue global loads whose latency the block then covers.
// The first TWO stages of a forward transform on decomposition digits in a field whose eighth roots of unity are short
// (FpG, FpH).  With E values per lane in layout A, slots (e, e + E/4, e + E/2, e + 3E/4) form a radix-4 group (d0, d2, d4,
// d6 below are those four, named for E = 8):
//   stage 1  a0 = d0 + b^2 d4, a4 = d0 - b^2 d4, (a2, a6 likewise from d2, d6)
//   stage 2  x0, x2 = a0 +- b a2;  x4, x6 = a4 +- b^3 a6
// and with b a2 = b d2 + b^3 d6, b^3 a6 = b^3 d2 - b^5 d6 = b^3 d2 + b d6 every term is digit x (at most 38 bits): exact,
// |x| <= 2^(logB + 38), no reduction.  Same residues as the general stages, much smaller representatives.
template <typename F, int M, int E>
__device__ __forceinline__ void fwd_top2_digits(double (&x)[M][E])
{
    constexpr int Q = E / 4;
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int e = 0; e < Q; e++) {
            const double d0 = x[m][e], d2 = x[m][e + Q], d4 = x[m][e + 2 * Q], d6 = x[m][e + 3 * Q];
            const double a0 = __builtin_fma(d4, F::B2, d0), a4 = __builtin_fma(d4, -F::B2, d0);
            const double u = __builtin_fma(d6, F::B3, d2 * F::B1), v = __builtin_fma(d6, F::B1, d2 * F::B3);
            x[m][e] = a0 + u;
            x[m][e + Q] = a0 - u;
            x[m][e + 2 * Q] = a4 + v;
            x[m][e + 3 * Q] = a4 - v;
            // every term is digit x (at most 38 bits): exact and far inside (-p/2, p/2) - the premise of the plain stages
            // (the 46-bit fields on 17-bit digits: exact below 2^51.5, 48 p of the 128 p a double holds - wide_headroom)
            [[maybe_unused]] constexpr double LIM = wide_headroom<F>::value ? 0x1p52 : F::P * 0.5;
            HELM_BOUND(__builtin_fabs(x[m][e]) < LIM && __builtin_fabs(x[m][e + Q]) < LIM &&
                           __builtin_fabs(x[m][e + 2 * Q]) < LIM && __builtin_fabs(x[m][e + 3 * Q]) < LIM, 2);
        }
}

// DIGITS (fields b^4 + 1 only): the inputs are decomposition digits -
//   2  |x| <= 2^12 and the table normalised to psi^(N/4) = b (the boolean engine): the first two stages as fwd_top2_digits;
//   1  |x| <= 2^23 (the 64-bit-torus engine): the first stage's products as plain multiplications by the table's psi^(N/2).
//   3  a HALF transform of a 2N-point transform on digits (k_pbs_tri10): the caller has done the 2N-point transform's first two
//      stages - the split into halves and this half's first stage (stride bit LOGN-1) - as plain products of digits with the
//      short roots, exactly the values fwd_top2_digits produces; the transform continues at stride bit LOGN-2.  The twiddle
//      source must hold the 2N-point table's entries of this half.  Bounds as DIGITS == 2 of the 2N-point transform (the same
//      butterflies): block A ends below 1.2 p, so SKIP_T1 applies.
template <typename F, int LOGN, int M, typename TW, int PRIO = 0, typename HOOK = NoHook, int DIGITS = 0>
__device__ __forceinline__ void ntt_forward(double (&x)[M][Geo<LOGN>::E], double *xbuf, const TW &tw, int lane,
                                            const HOOK &before_last = HOOK())
{
    using G = Geo<LOGN>;
    if constexpr (DIGITS == 1 && has_short_roots<F>::value)
        fwd_block<F, LOGN, M, 6, LOGN - 1, LOGN - G::BA, 0, TW, true>(x, tw, G::jA(lane, 0));
    else if constexpr (DIGITS == 2 && has_short_roots<F>::value) {
        static_assert(G::BA >= 3, "stages 1 and 2 pair slots e, e + E/2 and e, e + E/4 of block A");
        fwd_top2_digits<F, M, G::E>(x);
        fwd_block<F, LOGN, M, 6, LOGN - 3, LOGN - G::BA, 3>(x, tw, G::jA(lane, 0)); // the rest of block A
    } else if constexpr (DIGITS == 3 && has_short_roots<F>::value) {
        static_assert(G::BA >= 2, "the caller has done stride bit LOGN-1 of block A");
        fwd_block<F, LOGN, M, 6, LOGN - 2, LOGN - G::BA, 1>(x, tw, G::jA(lane, 0)); // the rest of block A
    } else
        fwd_block<F, LOGN, M, 6, LOGN - 1, LOGN - G::BA, 0>(x, tw, G::jA(lane, 0));
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO - 1);
    double *pA = xbuf + G::baseA(lane), *pB = xbuf + G::baseB(lane), *pC = xbuf + G::baseC(lane);
    // the recentring fields: after fwd_top2_digits block A ends below 1.2 p (its first two stages add almost nothing), so
    // block B (three stages) stays below 4.5 p of 2^53 = 5.26 p without a recentring here (tests/test_lazy_bounds.py)
    constexpr bool SKIP_T1 = (DIGITS == 2 || DIGITS == 3) && has_short_roots<F>::value && G::BB == 3;
    static_assert(!SKIP_T1 || (G::BA >= 3 && G::BB == 3), "SKIP_T1: block A ends below 1.2 p only behind fwd_top2_digits, and block B must be three stages");
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int e = 0; e < G::E; e++) pA[m * G::XPAD + G::offA1(e)] = SKIP_T1 ? x[m][e] : reduce_unless_lazy<F>(x[m][e]);
    lds_wave_sync();
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int e = 0; e < G::E; e++) x[m][e] = pB[m * G::XPAD + G::offB1(e)];
    lds_wave_sync();
    fwd_block<F, LOGN, M, G::BC, G::BC + G::BB - 1, G::BC, G::TWA>(x, tw, G::jB(lane, 0));
    if constexpr (PRIO > 1) __builtin_amdgcn_s_setprio(PRIO - 2);
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int e = 0; e < G::E; e++) pB[m * G::XPAD + G::offB2(e)] = reduce_unless_lazy<F>(x[m][e]);
    lds_wave_sync();
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int e = 0; e < G::E; e++) x[m][e] = pC[m * G::XPAD + e];
    lds_wave_sync();
    before_last();
    fwd_block<F, LOGN, M, 0, G::BC - 1, 0, G::TWA + G::TWB>(x, tw, G::jC(lane, 0));
}

// ntt_forward on decomposition digits (see DIGITS above).  -DHELM_SHORT_ROOT_STAGES=0 keeps the general stages (A/B).
#ifndef HELM_SHORT_ROOT_STAGES
#define HELM_SHORT_ROOT_STAGES 1
#endif
template <typename F, int LOGN, int M, int PRIO = 0, typename TW>
__device__ __forceinline__ void ntt_forward_digits(double (&x)[M][Geo<LOGN>::E], double *xbuf, const TW &tw, int lane)
{
    ntt_forward<F, LOGN, M, TW, PRIO, NoHook, HELM_SHORT_ROOT_STAGES != 0 ? 2 : 0>(x, xbuf, tw, lane);
}

// Inverse (without the 1/N factor, which is folded into the bootstrapping key).
// in : x[e] = transform word at jC(lane,e), |x| <= 0.5p.
// out: x[e] = coefficient jA(lane,e), exactly centred (|x| <= p/2).
// CENTRE = false leaves the outputs as the last block produced them (|x| <= 8 * 0.5 p after a three- or four-stage
// block: for callers that recentre downstream anyway).
// before_write: called after the first block, before the transform's first write into xbuf (a caller whose xbuf is
// still being read by another wave waits there instead of before the transform).
template <typename F, int LOGN, typename TW, int PRIO = 0, bool CENTRE = true, typename HOOK = NoHook>
__device__ __forceinline__ void ntt_inverse(double (&x)[Geo<LOGN>::E], double *xbuf, const TW &tw, int lane,
                                            const HOOK &before_write = HOOK())
{
    using G = Geo<LOGN>;
    // LEAN (lazy fields, three blocks of three stages, centred output): after a three-stage Gentleman-Sande block on inputs
    // <= m, slot e holds at most 8 m (e = 0: the pure sum), 4 x a product (e = 1), 2 x (e = 2, 3) or a fresh product
    // (e >= 4) - from m = p/2: 4.0, 2.4, 1.4, 1.3, 0.9, 0.8, 0.7, 0.6 p.  A transpose gives a lane eight values of ONE
    // slot class, so the next block's bound is eight times the largest value left unreduced: recentring slots 0-2 at
    // the first transpose (the rest <= 1.27 p) and 0-4 at the second (<= 0.8 p) keeps every sum below 0.75 x 2^53
    // (tests/test_lazy_bounds.py) and saves 8 of the 16 recentrings of the two transposes (24 instructions).
#ifndef HELM_LEAN_INVERSE
#define HELM_LEAN_INVERSE 1
#endif
    constexpr bool LEAN = HELM_LEAN_INVERSE != 0 && F::LAZY && lean_inverse_ok<F>::value && CENTRE && LOGN == 9 && G::BA == 3 && G::BB == 3 && G::BC == 3;
    // what the slot classes above assume of the layout: eight values per lane, three blocks of three stages, so that a
    // transpose hands a lane eight values of ONE slot of the block before (tests/test_lazy_bounds.py recomputes the bounds)
    static_assert(!LEAN || (G::E == 8 && G::BA + G::BB + G::BC == LOGN), "LEAN inverse: slot-class bounds are derived for Geo<9>");
    // WIDE (fields with 2^53 / p >= 128, three blocks of three stages, centred output; tests/test_lazy_bounds.py): inputs up to
    // 4.5 p UNREDUCED (a sum of four products of at most 1.1 p each).  After a three-stage block on inputs <= m the pure-sum slot holds 8 m, every other
    // slot passed a multiplication on the way (<= 2.3 p from m = 4 p); recentring slot 0 alone at each transpose keeps the next
    // block's inputs <= 2.6 p and every sum <= 8 x 4.5 p = 36 p, a quarter of what a double holds exactly.
    // (CENTRE = false: the outputs stay as the last block leaves them, <= 8 x 2.6 p = 21 p - for a caller that reduces downstream)
    constexpr bool WIDE = wide_headroom<F>::value && LOGN == 9 && G::BA == 3 && G::BB == 3 && G::BC == 3;
    if constexpr (LEAN) {
#pragma unroll
        for (int e = 0; e < G::E; e++) HELM_BOUND(__builtin_fabs(x[e]) <= F::P * 0.5000001, 3);
    }
    if constexpr (WIDE) {
#pragma unroll
        for (int e = 0; e < G::E; e++) HELM_BOUND(__builtin_fabs(x[e]) <= F::P * 4.5, 3);
    }
    inv_block<F, LOGN, 0, 0, G::BC - 1, 0>(x, tw, G::jC(lane, 0));
    before_write();
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO - 1);
    double *pA = xbuf + G::baseA(lane), *pB = xbuf + G::baseB(lane), *pC = xbuf + G::baseC(lane);
#pragma unroll
    for (int e = 0; e < G::E; e++) pC[e] = ((LEAN && e >= 3) || (WIDE && e >= 1)) ? x[e] : reduce<F>(x[e]);
    lds_wave_sync();
#pragma unroll
    for (int e = 0; e < G::E; e++) x[e] = pB[G::offB2(e)];
    lds_wave_sync();
    inv_block<F, LOGN, G::BC, G::BC, G::BC + G::BB - 1, G::TWC>(x, tw, G::jB(lane, 0));
    if constexpr (PRIO > 1) __builtin_amdgcn_s_setprio(PRIO - 2);
#pragma unroll
    for (int e = 0; e < G::E; e++) pB[G::offB1(e)] = ((LEAN && e >= 5) || (WIDE && e >= 1)) ? x[e] : reduce<F>(x[e]);
    lds_wave_sync();
#pragma unroll
    for (int e = 0; e < G::E; e++) x[e] = pA[G::offA1(e)];
    lds_wave_sync();
    inv_block<F, LOGN, 6, LOGN - G::BA, LOGN - 1, G::TWC + G::TWB>(x, tw, G::jA(lane, 0));
    if constexpr (CENTRE) {
#pragma unroll
        for (int e = 0; e < G::E; e++) x[e] = reduce<F>(x[e]);
    } else if constexpr (WIDE) {
        // nothing: see WIDE above
    } else if constexpr (G::BA == 4) {
        // four doublings from 0.5 p: the pure-sum slot reaches 8 p, every other slot passed a multiplication on the way
        // (<= 4.8 p); recentring slot 0 keeps sums and differences of two such values below 2^53 = 14.2 p
        static_assert(F::LAZY, "uncentred outputs need the headroom of the 49-bit fields");
        x[0] = reduce<F>(x[0]);
    }
}

} // namespace helm
