// helm_shortint.hip — gfx950 kernels + C ABI (include/helm_shortint.h) of the LUT-mode /
// arithmetic-mode engine: 64-bit torus, KS_PBS order.  Replaces the tfhe::shortint
// ServerKey calls HELM issues per LUT gate (reference src/gates.rs:754-785) and, through
// the radix layer built on the same two primitives, per arithmetic operator
// (src/gates.rs:306-702):
//
//   k_lincomb64     out = sum coef * in + const                      (no bootstrap)
//   k_keyswitch64   big LWE (k*N) -> small LWE (n), four ciphertexts per key pass
//   k_pbs64         modulus switch + blind rotate with a per-ciphertext look-up table +
//                   sample extract.  One workgroup per ciphertext, one wave per
//                   (accumulator polynomial, CRT prime): the exact negacyclic products of
//                   a 64-bit torus need ~2^97, i.e. two of the fp64 NTT fields of
//                   ntt_fp64.h and a CRT lift mod 2^64 after the inverse transforms.
//   k_bsk_convert64 standard-domain key -> both NTT fields (once per key)
//
// There is no CPU fallback in this file.
#include "../../include/helm_shortint.h"
#include "../../include/helm_comm.h"
#include "ntt_fp64.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#ifndef HELM_SI_TU
#define HELM_SI_TU 0 // 1: the max-ILP translation unit (two launchers only)
#endif
#ifndef HELM_SI_SPLIT_TU
#define HELM_SI_SPLIT_TU 0
#endif
using namespace helm;

int helm_hip_fail_(int code, const std::string &msg); // helm_hip.hip: sets helm_hip_last_error()
#define fail helm_hip_fail_
int helm_hip_runtime_guard_(const char *where); // helm_comm.cpp: one HIP runtime per process, or HELM_ERR_STATE naming the copies
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return fail(HELM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));          \
    } while (0)

namespace {

// CRT pair: two 49-bit primes whose 2^53 / p >= 13.3 headroom lets a whole forward transform (up to
// 11 stages, digits below 2^23: <= 9.6 p), the pointwise products and their sums (<= 11.4 p, the largest bound in this
// file) run without recentring.  Round 4: the pair is 5072^4 + 1 and 5096^4 + 1 (ntt_fp64.h FpG, FpG2; rounds 1-3:
// Fp<49>, Fp49b with 2^53 / p = 14.2) - their fourth root of unity psi^(N/2) = +-b^2 is 25 bits long, so stage 1 of a
// forward transform on digits (|d| <= 2^23: the split kernel is only chosen for pbs_logB <= 24) is ONE multiplication,
// exact and inside (-p/2, p/2), instead of a modular one (HELM_SI_PLAIN_STAGE1; p0 p1 / 2 = 2^97.5 covers more than before).
using F0 = FpG;
using F1 = FpG2;
// Round 6: the 46-bit pair (ntt_fp64.h FpJ, FpJ2) for k_pbs64k contexts whose LOADED key keeps the exact products below
// p p' / 2 = 2^90.62 (helm_si_load_bootstrap_key; helm_si_field_bits() says which pair a context computes in)
using J0 = FpJ;
using J1 = FpJ2;
#ifndef HELM_SI_PLAIN_STAGE1
#define HELM_SI_PLAIN_STAGE1 1
#endif
template <typename F>
__device__ __forceinline__ double stage1_digit_product(double digit, double w1)
{
    if constexpr (HELM_SI_PLAIN_STAGE1) return digit * w1; // |digit| <= 2^23, |w1| = b^2 < 2^24.7: exact, < p/2
    else return mulmod<F>(digit, w1);
}

#ifdef HELM_WIDE_STAMPS
__device__ unsigned long long g_stamps64[8 * 8];
#define STAMP_DECL unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1;
#define STAMP_BEGIN                        \
    t0 = __builtin_amdgcn_s_memtime();     \
    __builtin_amdgcn_s_waitcnt(0xC07F);
#define STAMP(k)                           \
    t1 = __builtin_amdgcn_s_memtime();     \
    __builtin_amdgcn_s_waitcnt(0xC07F);    \
    ph[k] += t1 - t0;                      \
    t0 = t1;
#define STAMP_END(wave)                    \
    if (blockIdx.x == 0 && lane == 0)      \
        for (int q = 0; q < 8; q++) g_stamps64[(wave) * 8 + q] = ph[q];
#else
#define STAMP_DECL
#define STAMP_BEGIN
#define STAMP(k)
#define STAMP_END(wave)
#endif

struct Pbs64Job {
    int32_t in_row;  // row of the small-LWE buffer (n+1 words)
    int32_t lut;     // row of the look-up-table buffer (N words)
    int32_t out_row; // row of the destination table (k*N+1 words)
    int32_t pad;
};

struct Ks64Job {
    int32_t in_row;  // row of the big table
    int32_t out_row; // row of the small-LWE buffer
};

__device__ __forceinline__ uint32_t modswitch64(uint64_t x, int log2_2N)
{
    uint32_t r = (uint32_t)(x >> (64 - log2_2N - 1)) + 1u;
    return (r >> 1) & ((1u << log2_2N) - 1u);
}

// exact integer in a double (|v| < 2^51) -> two's complement int64
__device__ __forceinline__ int64_t to_int64(double v)
{
    const double m = v + 6755399441055744.0; // 1.5 * 2^52: mantissa = 2^51 + v
    const uint32_t lo = (uint32_t)__double2loint(m);
    const int32_t hi = (int32_t)((uint32_t)__double2hiint(m) & 0xFFFFFu) - 0x80000;
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------
// k_pbs64<LOGN, L>: k = 1.  Waves w = 0..3: polynomial p = w >> 1, field f = w & 1.
// Per CMUX step, wave (p, f): rotate/subtract polynomial p (u64, LDS), signed-decompose,
// per level forward NTT in field f and multiply by the two key polynomials of row p; the
// product for the other polynomial goes to the partner (1-p, f) through this wave's
// scratch; inverse NTT of the own sum; the two residues of each coefficient meet in the
// wave that owns that half of the polynomial, which lifts them (CRT) to the exact integer
// mod 2^64 and accumulates.
// ------------------------------------------------------------------------------------
template <int LOGN_, int L_>
struct Pbs64Cfg {
    static constexpr int LOGN = LOGN_, L = L_, K = 1, K1 = 2, NW = 4;
    using G = Geo<LOGN>;
    static constexpr int MAX_SMALL_N = 1024;
    static constexpr size_t X_OFF = 0;                                            // double [NW][XPAD]
    // twiddles per field: index table for blocks A, B (N >> BC entries) + lane table for block C
    static constexpr int TW_IDX = G::N >> G::BC, TW_FIELD = TW_IDX + G::TWC * 64;
    static constexpr size_t TW_OFF = X_OFF + sizeof(double) * NW * G::XPAD;       // double [2][TW_FIELD]
    static constexpr size_t ACC_OFF = TW_OFF + sizeof(double) * 2 * TW_FIELD;     // u64 [K1][N]
    static constexpr size_t MS_OFF = ACC_OFF + sizeof(uint64_t) * K1 * G::N;      // u16 [n+1]
    static constexpr size_t BYTES = (MS_OFF + sizeof(uint16_t) * (MAX_SMALL_N + 1) + 15) / 16 * 16;
};

template <typename C, typename F, int MODE>
__device__ __forceinline__ void pbs64_body(unsigned char *smem, const double *__restrict__ bsk, int n, int logB,
                                           double p0inv_mod_p1, int p, int f, int lane,
                                           const uint64_t *__restrict__ alt_p)
{
    constexpr int LOGN = C::LOGN, L = C::L, K1 = C::K1;
    using G = Geo<LOGN>;
    constexpr int N = G::N, E = G::E, H = E / 2;
    double *X = reinterpret_cast<double *>(smem + C::X_OFF);
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    const uint16_t *MS = reinterpret_cast<const uint16_t *>(smem + C::MS_OFF);
    const int w = p * 2 + f;
    double *xb = X + (size_t)w * G::XPAD;                       // own scratch
    const double *x_poly = X + (size_t)((1 - p) * 2 + f) * G::XPAD; // same field, other polynomial
    const double *x_field = X + (size_t)(p * 2 + (1 - f)) * G::XPAD; // same polynomial, other field
    uint64_t *acc_p = ACC + (size_t)p * N;
    const double *twt = reinterpret_cast<const double *>(smem + C::TW_OFF) + (size_t)f * C::TW_FIELD;
    TwHybrid<LOGN, false> twf{twt, twt + C::TW_IDX + lane};
    TwHybrid<LOGN, true> twi{twt, twt + C::TW_IDX + (63 - lane)};

    // key words of step i for this wave: [i][row p][c][lev][f][e/2][lane] as double2
    const size_t per_poly = (size_t)2 * (N / 2); // both fields
    const size_t bsk_step = (size_t)K1 * K1 * L * per_poly;
    const double2 *bsk_w = reinterpret_cast<const double2 *>(bsk) + ((size_t)p * K1 * L * 2 + f) * (N / 2) + lane;

    STAMP_DECL
    for (int i = 0; i < n; i++) {
        const int a = __builtin_amdgcn_readfirstlane((int)MS[i]);
        if (a == 0) continue; // uniform over the workgroup: every wave skips the same steps
        STAMP_BEGIN
        const double2 *bp_i = bsk_w + (size_t)i * bsk_step;

        // ---- rotate / subtract, decomposition state (least significant level first; the
        //      rounded value has logB * L <= 32 bits) --------------------------------------------
        uint32_t state[E];
        {
            const int rep = logB * L;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int j = G::jA(lane, e);
                uint64_t v;
                if constexpr (MODE == 2) {
                    v = alt_p[j]; // CMUX of two given ciphertexts: c1 - c0
                } else {
                    const int src = (j - a) & (2 * N - 1);
                    v = acc_p[src & (N - 1)];
                    if (src >= N) v = 0ull - v;
                }
                v -= acc_p[j];
                state[e] = (uint32_t)((v + (1ull << (63 - rep))) >> (64 - rep));
            }
        }
        double mine[E], other[E];
        const uint32_t half_m1 = (1u << (logB - 1)) - 1u;
    [[maybe_unused]] const uint32_t bmask = (1u << logB) - 1u;
#pragma unroll
        for (int lev = L - 1; lev >= 0; lev--) {
            double x[1][E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                // tfhe's carry rule as one addition (see decompose_step in helm_hip.hip): the digit is what
                // the state loses when B/2 - 1 + (bit 2 logB - 1) is added and logB bits are shifted out
                const uint32_t s = state[e];
                const uint32_t next = (s + half_m1 + __builtin_amdgcn_ubfe(s, 2 * logB - 1, 1)) >> logB;
                state[e] = next;
                x[0][e] = (double)((int32_t)s - (int32_t)(next << logB));
            }
            // key words in chunks of E/4 (two per column), software-pipelined: the first chunk is
            // fetched before the transform, each next one just before the products of the previous
            // (at most two chunks = 64 registers in flight; whole columns spilled to AGPRs)
            constexpr int CH = E / 4; // double2 per chunk
            auto fetch = [&](int q, double2 (&dst)[CH]) { // chunk q of 4: column q / 2, half q % 2
                const double2 *kp = bp_i + (size_t)((q >> 1) * L + lev) * per_poly + (size_t)(q & 1) * CH * 64;
#pragma unroll
                for (int u = 0; u < CH; u++) dst[u] = kp[u * 64];
            };
            auto products = [&](int q, const double2 (&kq)[CH]) {
                const int c = q >> 1, e20 = (q & 1) * CH;
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int e2 = e20 + u;
                    const double t0 = mulmod<F>(x[0][2 * e2], kq[u].x), t1 = mulmod<F>(x[0][2 * e2 + 1], kq[u].y);
                    if (c == p) {
                        mine[2 * e2] = lev == L - 1 ? t0 : mine[2 * e2] + t0;
                        mine[2 * e2 + 1] = lev == L - 1 ? t1 : mine[2 * e2 + 1] + t1;
                    } else {
                        other[2 * e2] = lev == L - 1 ? t0 : other[2 * e2] + t0;
                        other[2 * e2 + 1] = lev == L - 1 ? t1 : other[2 * e2 + 1] + t1;
                    }
                }
            };
            double2 ka[CH], kb2[CH];
            fetch(0, ka);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0) // rotation, decomposition, first key chunk issued
            ntt_forward<F, LOGN, 1, decltype(twf), 0, NoHook, HELM_SI_PLAIN_STAGE1>(x, xb, twf, lane); // (digits: stage 1 plain)
            STAMP(1) // forward transform
            __builtin_amdgcn_sched_barrier(0);
            fetch(1, kb2);
            __builtin_amdgcn_sched_barrier(0);
            products(0, ka);
            __builtin_amdgcn_sched_barrier(0);
            fetch(2, ka);
            __builtin_amdgcn_sched_barrier(0);
            products(1, kb2);
            __builtin_amdgcn_sched_barrier(0);
            fetch(3, kb2);
            __builtin_amdgcn_sched_barrier(0);
            products(2, ka);
            products(3, kb2);
        }
        // hand the other polynomial's partial sum over through the (now idle) scratch
#pragma unroll
        for (int e = 0; e < E; e++) xb[e * 64 + lane] = reduce<F>(other[e]);
        STAMP(2) // products, hand-over written
        lds_block_sync();
        STAMP(3) // barrier 1
#pragma unroll
        for (int e = 0; e < E; e++) mine[e] = reduce<F>(reduce<F>(mine[e]) + x_poly[e * 64 + lane]);
        lds_block_sync(); // hand-over slots read: scratch free again
        STAMP(4) // sum + barrier 2

        ntt_inverse<F, LOGN>(mine, xb, twi, lane);
        STAMP(5) // inverse transform

        // ---- CRT: field-f wave lifts slots [f*H, f*H+H); it needs the other field's
        //      residues for those and provides its own for the other half ----------------
#pragma unroll
        for (int e = 0; e < H; e++) xb[e * 64 + lane] = mine[(1 - f) * H + e];
        lds_block_sync();
#pragma unroll
        for (int e = 0; e < H; e++) {
            const double own = mine[f * H + e], oth = x_field[e * 64 + lane];
            const double r0 = f == 0 ? own : oth, r1 = f == 0 ? oth : own;
            // x = r0 + p0 * t,  t = (r1 - r0) * p0^-1 mod p1 : the exact integer (|x| < p0 p1 / 2)
            const double t = mulmod<F1>(r1 - r0, p0inv_mod_p1);
            const uint64_t xv = (uint64_t)to_int64(r0) + F0::P_U64 * (uint64_t)to_int64(t);
            acc_p[G::jA(lane, f * H + e)] += xv;
        }
        lds_block_sync(); // accumulator complete before the next step's rotated reads
        STAMP(6) // CRT exchange, lift, accumulate, barriers 3 and 4
    }
    STAMP_END(p * 2 + f)
}

// MODE 0: programmable bootstrap (job: in_row of `small`, lut row of N words, out_row of k*N+1 words).
// MODE 1: blind rotation of a given GLWE with a per-job GGSW stack, sample extract - the tail of vertical packing
//         (job: in_row = row of `small` holding the synthetic rotation amounts, lut = row of (k+1) N words in
//         `luts`, pad = index of the job's key, key_stride doubles apart).
// MODE 2: one CMUX  out = c0 + GGSW (x) (c1 - c0)  (job: lut = row of c0, in_row = row of c1, both rows of
//         (k+1) N words of `luts`; pad = key index; the GGSW is entry key_first of that key; out rows of (k+1) N).
template <typename C, int MODE>
__global__ __launch_bounds__(64 * C::NW, 1) void k_pbs64(const Pbs64Job *__restrict__ jobs,
                                                         const uint64_t *__restrict__ small, // rows of n+1
                                                         const uint64_t *__restrict__ luts,  // rows of N
                                                         const double *__restrict__ bsk,     // NTT domain, both fields
                                                         const double *__restrict__ tw0,
                                                         const double *__restrict__ tw1,
                                                         uint64_t *__restrict__ out, // rows of k*N+1
                                                         int n, int logB, double p0inv_mod_p1, size_t key_stride,
                                                         int key_first)
{
    constexpr int LOGN = C::LOGN, K = C::K;
    using G = Geo<LOGN>;
    constexpr int N = G::N, E = G::E, H = E / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    uint16_t *MS = reinterpret_cast<uint16_t *>(smem + C::MS_OFF);
    double *TW = reinterpret_cast<double *>(smem + C::TW_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = w >> 1, f = w & 1;
    const Pbs64Job job = jobs[blockIdx.x];
    if constexpr (MODE == 2) {
        if (tid == 0) MS[0] = 1, MS[1] = 0;
    } else {
        const uint64_t *lwe = small + (size_t)job.in_row * ((size_t)n + 1);
        for (int i = tid; i <= n; i += 64 * C::NW) MS[i] = (uint16_t)modswitch64(lwe[i], LOGN + 1);
    }
    for (int i = tid; i < C::TW_IDX; i += 64 * C::NW) {
        TW[i] = tw0[i];
        TW[C::TW_FIELD + i] = tw1[i];
    }
    for (int r = w; r < G::TWC; r += C::NW) { // lane-table rows of block C (slots TWB.. of tw_lane_index)
        const int idx = tw_lane_index<LOGN>(G::TWB + r, lane);
        TW[C::TW_IDX + r * 64 + lane] = tw0[idx];
        TW[C::TW_FIELD + C::TW_IDX + r * 64 + lane] = tw1[idx];
    }
    __syncthreads();
    // accumulator: (0, X^{-b~} * lut); MODE 1, 2: the given GLWE as it is
    if constexpr (MODE == 0) {
        const int bt = (int)MS[n];
        const uint64_t *tv = luts + (size_t)job.lut * N;
        for (int j = tid; j < (K + 1) * N; j += 64 * C::NW) {
            uint64_t v = 0;
            if (j >= K * N) {
                const int idx = ((j - K * N) + bt) & (2 * N - 1);
                v = tv[idx & (N - 1)];
                if (idx >= N) v = 0ull - v;
            }
            ACC[j] = v;
        }
    } else {
        const uint64_t *c0 = luts + (size_t)job.lut * ((size_t)(K + 1) * N);
        for (int j = tid; j < (K + 1) * N; j += 64 * C::NW) ACC[j] = c0[j];
    }
    __syncthreads();

    const double *key = bsk;
    const uint64_t *alt_p = nullptr;
    if constexpr (MODE != 0) key += (size_t)job.pad * key_stride + (size_t)key_first * ((size_t)(K + 1) * (K + 1) * C::L * 2 * N);
    if constexpr (MODE == 2) alt_p = luts + (size_t)job.in_row * ((size_t)(K + 1) * N) + (size_t)p * N;
    const int steps = MODE == 2 ? 1 : n;
    if (f == 0) pbs64_body<C, F0, MODE>(smem, key, steps, logB, p0inv_mod_p1, p, 0, lane, alt_p);
    else pbs64_body<C, F1, MODE>(smem, key, steps, logB, p0inv_mod_p1, p, 1, lane, alt_p);

    const uint64_t *acc_p = ACC + (size_t)p * N;
    if constexpr (MODE == 2) { // the whole GLWE
        uint64_t *og = out + (size_t)job.out_row * ((size_t)(K + 1) * N) + (size_t)p * N;
#pragma unroll
        for (int e = 0; e < H; e++) {
            const int j = G::jA(lane, f * H + e);
            og[j] = acc_p[j];
        }
        return;
    }
    // ---- sample extract (coefficient 0); wave (p, f) writes its half of the slots ------
    uint64_t *ob = out + (size_t)job.out_row * ((size_t)K * N + 1);
    if (p < K) {
#pragma unroll
        for (int e = 0; e < H; e++) {
            const int j = G::jA(lane, f * H + e);
            const uint64_t v = acc_p[j];
            if (j == 0) ob[p * N] = v;
            else ob[p * N + (N - j)] = 0ull - v;
        }
    } else if (f == 0 && lane == 0) {
        ob[K * N] = acc_p[0];
    }
}

// ------------------------------------------------------------------------------------
// k_pbs64k<LOGN, K>: the programmable bootstrap for k > 1 and one decomposition level - the
// parameter set the reference binary installs for LUT mode, PARAM_MESSAGE_1_CARRY_1_KS_PBS
// (reference src/bin/helm.rs:301: k = 3, N = 512 [dimensions recalled]).  2 (k+1) waves:
// wave w = (polynomial p = w >> 1, field f = w & 1), as k_pbs64; what changes with k + 1 > 2
// polynomials is the hand-over: a wave's product with key column c belongs to polynomial c, and
// the k foreign ones are summed into the (cleared) transform scratch of wave (c, f) with ds_add_f64 -
// exact integers below 2^53, so the sum does not depend on the order the waves arrive in.  Four
// workgroup barriers per step; 64 KB of LDS and at most 128 registers: TWO ciphertexts per CU.
// ------------------------------------------------------------------------------------
#ifndef HELM_SI_K_SPLIT_DIGITS
#define HELM_SI_K_SPLIT_DIGITS 1
#endif
#ifndef HELM_SI_K_ACC2
#define HELM_SI_K_ACC2 1 // k_pbs64k: the accumulator polynomials stored with their negated copies behind them
#endif
#ifndef HELM_SI_K_GATHER
#define HELM_SI_K_GATHER 1 // k_pbs64k: column sums gathered by the owner of the column instead of scattered with ds_add_f64
#endif
template <int LOGN_, int K_>
struct Pbs64kCfg {
    static constexpr int LOGN = LOGN_, L = 1, K = K_, K1 = K_ + 1, NW = 2 * K1;
    using G = Geo<LOGN>;
    static constexpr int MAX_SMALL_N = 1024;
    static constexpr size_t X_OFF = 0;                                            // double [NW][XPAD]: transform scratch,
                                                                                  // and between two barriers the column sums
    static constexpr int TW_IDX = G::N >> G::BC, TW_FIELD = TW_IDX + G::TWC * 64;
    static constexpr size_t TW_OFF = X_OFF + sizeof(double) * NW * G::XPAD;       // double [2][TW_FIELD]
    static constexpr size_t ACC_OFF = TW_OFF + sizeof(double) * 2 * TW_FIELD;     // u64 [K1][N]
    // HELM_SI_K_ACC2: every polynomial is stored as [acc | -acc] (2N words): a rotated read X^a acc is then ONE indexed read,
    // no sign logic (5 vector instructions per coefficient in a kernel that is bound by its instruction stream)
    static constexpr int ACC_LEN = HELM_SI_K_ACC2 ? 2 * G::N : G::N;
    static constexpr size_t MS_OFF = ACC_OFF + sizeof(uint64_t) * K1 * ACC_LEN;   // u16 [n+1]
    static constexpr size_t BYTES = (MS_OFF + sizeof(uint16_t) * (MAX_SMALL_N + 1) + 15) / 16 * 16;
    static_assert(NW <= 16, "a workgroup holds at most 16 waves");
    static_assert(2 * BYTES <= 160 * 1024, "two ciphertexts per CU");
};

// F: this wave's field; FA, FB: the CRT pair (F is one of them)
template <typename C, typename F, typename FA, typename FB>
__device__ __forceinline__ void pbs64k_body(unsigned char *smem, const double *__restrict__ bsk, int n, int logB,
                                            double p0inv_mod_p1, int p, int f, int lane)
{
    constexpr int LOGN = C::LOGN, K1 = C::K1;
    // 46-bit pair: both leading forward stages plain on the digits (|d| <= 2^17), no recentring of the column sums - the
    // inverse transform takes them unreduced (ntt_inverse, WIDE)
    constexpr bool WIDE = wide_headroom<F>::value;
    constexpr int DIG = WIDE ? 2 : HELM_SI_PLAIN_STAGE1;
    using G = Geo<LOGN>;
    constexpr int N = G::N, E = G::E, H = E / 2;
    double *X = reinterpret_cast<double *>(smem + C::X_OFF);
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    const uint16_t *MS = reinterpret_cast<const uint16_t *>(smem + C::MS_OFF);
    double *xb = X + (size_t)(p * 2 + f) * G::XPAD;                          // own scratch
    const double *x_field = X + (size_t)(p * 2 + (1 - f)) * G::XPAD;         // same polynomial, other field
    uint64_t *acc_p = ACC + (size_t)p * C::ACC_LEN;
    const double *twt = reinterpret_cast<const double *>(smem + C::TW_OFF) + (size_t)f * C::TW_FIELD;
    TwHybrid<LOGN, false> twf{twt, twt + C::TW_IDX + lane};
    TwHybrid<LOGN, true> twi{twt, twt + C::TW_IDX + (63 - lane)};
    // key words of step i for this wave: [i][row p][c][f][e/2][lane] as double2 (the layout k_bsk_convert64 writes)
    // read with buffer loads: one descriptor in scalar registers, a scalar byte offset per (step, column), one lane
    // register and an immediate per word (plain pointers cost an address register pair per word and spill at 128)
    const unsigned poly_bytes = (unsigned)(N / 2) * 16u;      // one key polynomial in one field
    const unsigned col_bytes = 2u * poly_bytes;               // both fields
    const unsigned step_bytes = (unsigned)(K1 * K1) * col_bytes;
    const unsigned row_off = (HELM_SI_K_GATHER ? (unsigned)p : (unsigned)(p * K1)) * col_bytes + (unsigned)f * poly_bytes;
    KeyBuf kb;
    kb.init(bsk, (size_t)n * step_bytes, lane);
    const uint32_t half_m1 = (1u << (logB - 1)) - 1u;
    [[maybe_unused]] const uint32_t bmask = (1u << logB) - 1u;
    for (int i = 0; i < n; i++) {
        const int a = __builtin_amdgcn_readfirstlane((int)MS[i]);
        if (a == 0) continue; // uniform over the workgroup
        const unsigned so_i = (unsigned)i * step_bytes + row_off;
#if HELM_SI_K_GATHER
        // Gather form of the hand-over: wave (p, f) owns OUTPUT column p.  Every wave publishes the spectrum of its digit
        // polynomial in its scratch; after the barrier each wave reads all K1 spectra and multiplies them with the key
        // words of column p (rows 0..k): the column sum stays in registers - no LDS atomics, no clear, no data-dependent
        // branch, and the own spectrum's registers are free during the products.
        double2 kw[H][K1];
        auto fetch = [&](int u) {
#pragma unroll
            for (int r = 0; r < K1; r++) kw[u][r] = kb.load(so_i + (unsigned)(r * K1) * col_bytes, u * 1024);
        };
#pragma unroll
        for (int u = 0; u < H / 2; u++) fetch(u);
        double mine[E];
        {
            double x[1][E];
            auto digit = [&](int e) {
                const int j = G::jA(lane, e);
                const int src = (j - a) & (2 * N - 1);
                uint64_t v;
                if constexpr (HELM_SI_K_ACC2) v = acc_p[src];
                else {
                    v = acc_p[src & (N - 1)];
                    if (src >= N) v = 0ull - v;
                }
                v -= acc_p[j];
                const uint32_t st = (uint32_t)((v + (1ull << (63 - logB))) >> (64 - logB));
                return (double)((int)((st + half_m1) & bmask) - (int)half_m1); // st <= B/2 stays, above it st - B
            };
#if HELM_SI_K_SPLIT_DIGITS
            // the two field waves of a polynomial need the same digits: each makes half of them and hands them to the other
            // through the other's (free) transform scratch - one more barrier, half the decomposition work
            double *xb_field = X + (size_t)(p * 2 + (1 - f)) * G::XPAD;
#pragma unroll
            for (int e = 0; e < H; e++) {
                x[0][f * H + e] = digit(f * H + e);
                xb_field[(f * H + e) * 64 + lane] = x[0][f * H + e];
            }
            lds_block_sync();
#pragma unroll
            for (int e = 0; e < H; e++) x[0][(1 - f) * H + e] = xb[((1 - f) * H + e) * 64 + lane];
#else
#pragma unroll
            for (int e = 0; e < E; e++) x[0][e] = digit(e);
#endif
            ntt_forward<F, LOGN, 1, decltype(twf), 0, NoHook, DIG>(x, xb, twf, lane); // (digits: stage 1 - 46-bit pair: stages 1 and 2 - plain)
#pragma unroll
            for (int e = 0; e < E; e++) xb[e * 64 + lane] = x[0][e];
        }
        lds_block_sync(); // every spectrum published
        const double *xs = X + (size_t)f * G::XPAD + lane; // spectrum of polynomial r: xs + r * 2 * XPAD
        auto products = [&](int u) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int r = 0; r < K1; r++) { // |t| <= 1.5 p each: the sum of K1 <= 4 stays below 6 p < 2^53, exact
                const double *xr = xs + (size_t)r * 2 * G::XPAD;
                const double t0 = mulmod<F>(xr[(2 * u) * 64], kw[u][r].x), t1 = mulmod<F>(xr[(2 * u + 1) * 64], kw[u][r].y);
                s0 = r == 0 ? t0 : s0 + t0;
                s1 = r == 0 ? t1 : s1 + t1;
            }
            mine[2 * u] = WIDE ? s0 : reduce<F>(s0);   // (WIDE: |s| <= 4.5 p, what ntt_inverse's WIDE form takes)
            mine[2 * u + 1] = WIDE ? s1 : reduce<F>(s1);
        };
#pragma unroll
        for (int u = 0; u < H; u++) {
            if (u + H / 2 < H) {
                fetch(u + H / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            products(u);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_block_sync(); // every wave has read: the scratches are free for the inverse transforms
#else
        // key words: all K1 columns of the first half of the spectrum slots at the top (the transform covers them), the
        // second half slot pair by slot pair around the products (at most three quarters of the words live at once:
        // the kernel must stay within 128 registers for two ciphertexts per CU)
        double2 kw[H][K1];
        auto fetch = [&](int u) {
#pragma unroll
            for (int c = 0; c < K1; c++) kw[u][c] = kb.load(so_i + (unsigned)c * col_bytes, u * 1024);
        };
#pragma unroll
        for (int u = 0; u < H / 2; u++) fetch(u);
        // ---- rotate / subtract, one signed digit per coefficient (pbs_l = 1) ------------------
        double x[1][E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int j = G::jA(lane, e);
            const int src = (j - a) & (2 * N - 1);
            uint64_t v;
            if constexpr (HELM_SI_K_ACC2) v = acc_p[src];
            else {
                v = acc_p[src & (N - 1)];
                if (src >= N) v = 0ull - v;
            }
            v -= acc_p[j];
            const uint32_t st = (uint32_t)((v + (1ull << (63 - logB))) >> (64 - logB));
            x[0][e] = (double)((int)((st + half_m1) & bmask) - (int)half_m1); // st <= B/2 stays, above it st - B
        }
        ntt_forward<F, LOGN, 1, decltype(twf), 0, NoHook, DIG>(x, xb, twf, lane); // (digits: stage 1 - 46-bit pair: stages 1 and 2 - plain)
        // the scratch becomes this wave's column sum: clear it, and wait until every wave is through its transform
#pragma unroll
        for (int e = 0; e < E; e++) xb[e * 64 + lane] = 0.0;
        lds_block_sync();
        // ---- products: column p stays, the others go to the waves of their polynomials --------
        double mine[E];
        auto products = [&](int u) {
#pragma unroll
            for (int c = 0; c < K1; c++) {
                double *sum_c = X + (size_t)(c * 2 + f) * G::XPAD + lane;
                const double t0 = mulmod<F>(x[0][2 * u], kw[u][c].x), t1 = mulmod<F>(x[0][2 * u + 1], kw[u][c].y);
                if (c == p) {
                    mine[2 * u] = t0;
                    mine[2 * u + 1] = t1;
                } else { // |t| <= 1.5 p, k of them and the own one: <= 6 p < 2^53, every partial sum exact
                    lds_add_wg(sum_c + (2 * u) * 64, t0);
                    lds_add_wg(sum_c + (2 * u + 1) * 64, t1);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < H; u++) {
            if (u + H / 2 < H) {
                fetch(u + H / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            products(u);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_block_sync(); // every foreign product is in
#pragma unroll
        for (int e = 0; e < E; e++) mine[e] = reduce<F>(mine[e] + xb[e * 64 + lane]);
#endif
        // WIDE: the outputs stay unreduced (<= 21 p): the CRT below reduces the DIFFERENCE of the two residues once instead of
        // both residues (X = r0 + p0 t equals the exact integer for any representative r0 with |r0| + |V| + p0 |t| < p0 p1)
        ntt_inverse<F, LOGN, decltype(twi), 0, !WIDE>(mine, xb, twi, lane);
        // ---- CRT: field-f wave lifts slots [f*H, f*H+H) of its polynomial ---------------------
#pragma unroll
        for (int e = 0; e < H; e++) xb[e * 64 + lane] = mine[(1 - f) * H + e];
        lds_block_sync();
#pragma unroll
        for (int e = 0; e < H; e++) {
            const double own = mine[f * H + e], oth = x_field[e * 64 + lane];
            const double r0 = f == 0 ? own : oth, r1 = f == 0 ? oth : own;
            const double t = mulmod<FB>(WIDE ? reduce<FB>(r1 - r0) : r1 - r0, p0inv_mod_p1);
            const uint64_t xv = (uint64_t)to_int64(r0) + FA::P_U64 * (uint64_t)to_int64(t);
            const int j = G::jA(lane, f * H + e);
            const uint64_t nv = acc_p[j] + xv;
            acc_p[j] = nv;
            if constexpr (HELM_SI_K_ACC2) acc_p[j + N] = 0ull - nv;
        }
        lds_block_sync(); // accumulator complete before the next step's rotated reads; scratch free again
    }
}

template <typename C, typename FA, typename FB>
__global__ __launch_bounds__(64 * C::NW, C::NW / 2) void k_pbs64k(const Pbs64Job *__restrict__ jobs,
                                                          const uint64_t *__restrict__ small, // rows of n+1
                                                          const uint64_t *__restrict__ luts,  // rows of N
                                                          const double *__restrict__ bsk,     // NTT domain, both fields
                                                          const double *__restrict__ tw0, const double *__restrict__ tw1,
                                                          uint64_t *__restrict__ out, // rows of k*N+1
                                                          int n, int logB, double p0inv_mod_p1)
{
    constexpr int LOGN = C::LOGN, K = C::K;
    using G = Geo<LOGN>;
    constexpr int N = G::N, E = G::E, H = E / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    uint16_t *MS = reinterpret_cast<uint16_t *>(smem + C::MS_OFF);
    double *TW = reinterpret_cast<double *>(smem + C::TW_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = w >> 1, f = w & 1;
    const Pbs64Job job = jobs[blockIdx.x];
    const uint64_t *lwe = small + (size_t)job.in_row * ((size_t)n + 1);
    for (int i = tid; i <= n; i += 64 * C::NW) MS[i] = (uint16_t)modswitch64(lwe[i], LOGN + 1);
    for (int i = tid; i < C::TW_IDX; i += 64 * C::NW) {
        TW[i] = tw0[i];
        TW[C::TW_FIELD + i] = tw1[i];
    }
    for (int r = w; r < G::TWC; r += C::NW) { // lane-table rows of block C
        const int idx = tw_lane_index<LOGN>(G::TWB + r, lane);
        TW[C::TW_IDX + r * 64 + lane] = tw0[idx];
        TW[C::TW_FIELD + C::TW_IDX + r * 64 + lane] = tw1[idx];
    }
    __syncthreads();
    { // accumulator: (0, ..., 0, X^{-b~} * lut)
        const int bt = (int)MS[n];
        const uint64_t *tv = luts + (size_t)job.lut * N;
        for (int j = tid; j < (K + 1) * N; j += 64 * C::NW) {
            uint64_t v = 0;
            if (j >= K * N) {
                const int idx = ((j - K * N) + bt) & (2 * N - 1);
                v = tv[idx & (N - 1)];
                if (idx >= N) v = 0ull - v;
            }
            const int pp = j / N, jj = j - pp * N;
            ACC[(size_t)pp * C::ACC_LEN + jj] = v;
            if constexpr (HELM_SI_K_ACC2) ACC[(size_t)pp * C::ACC_LEN + N + jj] = 0ull - v;
        }
    }
    __syncthreads();
    if (f == 0) pbs64k_body<C, FA, FA, FB>(smem, bsk, n, logB, p0inv_mod_p1, p, 0, lane);
    else pbs64k_body<C, FB, FA, FB>(smem, bsk, n, logB, p0inv_mod_p1, p, 1, lane);
    // ---- sample extract (coefficient 0); wave (p, f) writes its half of the slots ------
    const uint64_t *acc_p = ACC + (size_t)p * C::ACC_LEN;
    uint64_t *ob = out + (size_t)job.out_row * ((size_t)K * N + 1);
    if (p < K) {
#pragma unroll
        for (int e = 0; e < H; e++) {
            const int j = G::jA(lane, f * H + e);
            const uint64_t v = acc_p[j];
            if (j == 0) ob[p * N] = v;
            else ob[p * N + (N - j)] = 0ull - v;
        }
    } else if (f == 0 && lane == 0) {
        ob[K * N] = acc_p[0];
    }
}

// ------------------------------------------------------------------------------------
// k_pbs64s<LOGN>: the same bootstrap with EIGHT waves per ciphertext (two per SIMD), L = 1:
// wave w = (polynomial p, field f, half h).  A size-N negacyclic transform splits, after its
// first butterfly stage (pairs j, j + N/2, twiddle psi^(N/2)), into two independent size-N/2
// transforms whose twiddles are slices of the full table: half h, stage with m' groups, group i'
// uses table[2m' + h m' + i'].  So wave (p, f, h) transforms half h with the LOGN-1 machinery of
// ntt_fp64.h on a derived table; by the mirror identity the inverse of half h reads the table of
// half 1-h mirrored.  The halves meet once per step, in the last inverse stage.
// Per step: (1) each of the four waves of a polynomial decomposes a quarter of its coefficients
// and publishes the digits (int32, LDS); (2) stage 1 + half transform + products with the key
// words of its half of the spectrum; other-polynomial sum handed over; (3) half inverse, exchange
// with the other half, last stage; (4) CRT with the other field on half of the wave's
// coefficients, accumulate.  Seven workgroup barriers per step, no redundant work except the
// 16 stage-1 products per lane.
// ------------------------------------------------------------------------------------
// LDS flags between two waves of a workgroup (LDS-only fences, as lds_block_sync: global loads stay in flight).
// set: everything this wave wrote to or read from LDS before is done when the value shows; wait: nothing this wave does
// to LDS afterwards starts before the value was seen.
[[maybe_unused]] __device__ __forceinline__ void lds_flag_set(uint32_t *flag, uint32_t v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
[[maybe_unused]] __device__ __forceinline__ void lds_flag_wait(const uint32_t *flag, uint32_t v)
{
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != v) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifndef HELM_SI_MB_NESTED
#define HELM_SI_MB_NESTED 1 // multi-bit: the group's key sum in nested form (2^g - 1 multiplications per position and column)
#endif
// the half transforms of k_pbs64s (a one-transpose form of the 1,024-point half - lane-bit stages through row swaps - was
// bit-identical and measured -0.4 % / +0.7 %: profiles/r03/si_kernel_experiments.txt; removed in round 6)
template <typename F, int LOGH, typename TW, int PRIO, typename HOOK = NoHook>
__device__ __forceinline__ void half_forward(double (&x)[1][Geo<LOGH>::E], double *xb, const TW &tw, int lane,
                                             const HOOK &hook = HOOK())
{
    ntt_forward<F, LOGH, 1, TW, PRIO, HOOK>(x, xb, tw, lane, hook);
}
template <typename F, int LOGH, typename TW, int PRIO, bool CENTRE, typename HOOK = NoHook>
__device__ __forceinline__ void half_inverse(double (&x)[Geo<LOGH>::E], double *xb, const TW &tw, int lane,
                                             const HOOK &hook = HOOK())
{
    ntt_inverse<F, LOGH, TW, PRIO, CENTRE, HOOK>(x, xb, tw, lane, hook);
}

template <int LOGN_, int L_ = 1>
struct Pbs64sCfg {
    static constexpr int LOGN = LOGN_, L = L_, K = 1, K1 = 2, NW = 8;
    using G = Geo<LOGN>;      // decomposition geometry: E coefficients per lane
    using GS = Geo<LOGN - 1>; // half transforms
    static constexpr int MAX_SMALL_N = 1024;
#ifndef HELM_SI_PRIO
#define HELM_SI_PRIO 1
#endif
#ifndef HELM_SI_FUSED_XCHG
#define HELM_SI_FUSED_XCHG 1 // one exchange for the last inverse stage and the CRT (two barriers instead of four)
#endif
#ifndef HELM_SI_KW1_EARLY
#define HELM_SI_KW1_EARLY 1 // second key column fetched before the LAST block of the forward half transform
#endif
#ifndef HELM_SI_STAGE1_SRC
#define HELM_SI_STAGE1_SRC 1 // k_pbs64s, one level: stage 1 of the full transform done by the wave that makes the digits
#endif
#ifndef HELM_SI_TWC_REGS
#define HELM_SI_TWC_REGS 1 // classical k_pbs64s, one level: the lane's block-C twiddles of both half transforms in registers
#endif
#ifndef HELM_SI_PAIR_LIFT
#define HELM_SI_PAIR_LIFT 1 // k_pbs64s: a wave lifts both outputs j and j + N/2 of a quarter of the slots (lift_pairs)
#endif
#ifndef HELM_SI_MIX_HALVES
#define HELM_SI_MIX_HALVES 1 // k_pbs64s: one wave of either transform half per SIMD (the halves' last stages differ in cost)
#endif
#ifndef HELM_SI_STATIC_P
#define HELM_SI_STATIC_P 1 // k_pbs64s: one inlined body per polynomial as well (the wave's polynomial is a literal inside)
#endif
#ifndef HELM_SI_LAZY_INV
#define HELM_SI_LAZY_INV 1 // the half inverse leaves its outputs uncentred: the last stage recentres anyway
#endif
    static constexpr bool PRIO = HELM_SI_PRIO != 0;
    static constexpr int TW_IDX = GS::N >> GS::BC, TW_PART = TW_IDX + GS::TWC * 64; // per (field, half)
    static constexpr size_t X_OFF = 0;                                              // double [NW][GS::XPAD]
    static constexpr size_t TW_OFF = X_OFF + sizeof(double) * NW * GS::XPAD;        // double [2][2][TW_PART]
    static constexpr size_t ACC_OFF = TW_OFF + sizeof(double) * 4 * TW_PART;        // u64 [K1][N]
    // digits: int32 for one level (23-bit digits), int16 for two (pbs_logB <= 15: |digit| <= 2^14) - the LDS of a CU
    // does not hold two levels of int32 next to the rest at N = 2048
    using dig_t = std::conditional_t<L == 1, int32_t, int16_t>;
    static constexpr size_t DIG_OFF = ACC_OFF + sizeof(uint64_t) * K1 * G::N;       // dig_t [K1][L][N]
    static constexpr size_t MS_OFF = DIG_OFF + sizeof(dig_t) * K1 * L * G::N;       // u16 [n+1]
    static constexpr size_t FLAG_OFF = (MS_OFF + sizeof(uint16_t) * (MAX_SMALL_N + 1) + 15) / 16 * 16; // u32 [2][NW]
    static constexpr size_t BYTES = FLAG_OFF + sizeof(uint32_t) * 2 * NW;
};

// The last inverse stage and the CRT of k_pbs64s, lifted in PAIRS (HELM_SI_PAIR_LIFT): wave (f, h) owns slots
// [(2 f + h) E/4, + E/4) of BOTH halves of its polynomial - from its own value of the slot and the three others' (other
// half, other field, other field's other half) it computes the last stage of both outputs j and j + N/2 in both fields and
// lifts both.  Against "every wave lifts the slots of its own half": 12 LDS reads per wave instead of 24 (and 12 values
// published instead of 16) in the phase that is bound by the LDS pipe, and the sixteen modular multiplications of the
// (a1 - a0) psi^(N/2) outputs spread evenly over the waves instead of sitting on the h = 1 ones.  Same values.
template <typename C, typename F, int h, bool ADD>
__device__ __forceinline__ void lift_pairs(const double (&mine)[C::GS::E], const double *x_half, const double *x_field,
                                           const double *x_fh, uint64_t *acc_p, int f, double w1, double w1o,
                                           double p0inv_mod_p1, int lane)
{
    using FO = std::conditional_t<std::is_same<F, F0>::value, F1, F0>;
    constexpr int EH = C::GS::E, QS = EH / 4, N = C::G::N;
    const int s0 = (f * 2 + h) * QS;
#pragma unroll
    for (int e = 0; e < QS; e++) {
        const int s = s0 + e;
        const double a = mine[s], o = x_half[s * 64 + lane], a2 = x_field[s * 64 + lane], o2 = x_fh[s * 64 + lane];
        const double e0 = h ? o : a, e1 = h ? a : o, g0 = h ? o2 : a2, g1 = h ? a2 : o2; // halves 0 and 1, own / other field
        // y[j] = a0 + a1 ; y[j + N/2] = (a1 - a0) * psi^(N/2)
        const double own[2] = {reduce<F>(e0 + e1), reduce<F>(mulmod<F>(e1 - e0, w1))};
        const double oth[2] = {reduce<FO>(g0 + g1), reduce<FO>(mulmod<FO>(g1 - g0, w1o))};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const double r0 = f == 0 ? own[u] : oth[u], r1 = f == 0 ? oth[u] : own[u];
            const double t = mulmod<F1>(r1 - r0, p0inv_mod_p1);
            const uint64_t xv = (uint64_t)to_int64(r0) + F0::P_U64 * (uint64_t)to_int64(t);
            uint64_t *dst = acc_p + u * (N / 2) + s * 64 + lane;
            *dst = ADD ? *dst + xv : xv;
        }
    }
}

template <typename C, typename F, int h, int P = -1>
__device__ __forceinline__ void pbs64s_body(unsigned char *smem, const double *__restrict__ bsk, int n, int logB,
                                            double p0inv_mod_p1, double w1, double w1o, int p, int f, int lane)
{
    constexpr int LOGN = C::LOGN, K1 = C::K1, L = C::L;
    using G = typename C::G;
    using GS = typename C::GS;
    constexpr int N = G::N, E = G::E, EH = GS::E, Q = E / 4, HC = EH / 2;
    // entry priority of the transforms (stepped down block by block inside them)
    constexpr int PH = !C::PRIO ? 0 : 3;
    constexpr bool SRC1 = HELM_SI_STAGE1_SRC && L == 1; // stage 1 where the digits are made
    double *X = reinterpret_cast<double *>(smem + C::X_OFF);
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    using dig_t = typename C::dig_t;
    dig_t *DIG = reinterpret_cast<dig_t *>(smem + C::DIG_OFF);
    const uint16_t *MS = reinterpret_cast<const uint16_t *>(smem + C::MS_OFF);
    auto wave_of = [](int pp, int ff, int hh) { return (pp * 2 + ff) * 2 + hh; };
    double *xb = X + (size_t)wave_of(p, f, h) * GS::XPAD;
    const double *x_poly = X + (size_t)wave_of(1 - p, f, h) * GS::XPAD;  // same field and half, other polynomial
    const double *x_half = X + (size_t)wave_of(p, f, 1 - h) * GS::XPAD;  // same polynomial and field, other half
    const double *x_field = X + (size_t)wave_of(p, 1 - f, h) * GS::XPAD; // same polynomial and half, other field
    uint64_t *acc_p = ACC + (size_t)p * N;
    dig_t *dig_p = DIG + (size_t)p * L * N; // [level][N]
    const double *twt = reinterpret_cast<const double *>(smem + C::TW_OFF);
    const double *tw_own = twt + (size_t)(f * 2 + h) * C::TW_PART, *tw_oth = twt + (size_t)(f * 2 + (1 - h)) * C::TW_PART;
    // block-C twiddles of both transforms in registers where the kernel has them to spare (one level, classical: 176 + 60)
    constexpr bool TWC_REGS = HELM_SI_TWC_REGS && L == 1;
    std::conditional_t<TWC_REGS, TwHybridC<LOGN - 1, false>, TwHybrid<LOGN - 1, false>> twf;
    std::conditional_t<TWC_REGS, TwHybridC<LOGN - 1, true>, TwHybrid<LOGN - 1, true>> twi;
    twf.t = tw_own, twi.t = tw_oth;
    if constexpr (TWC_REGS) {
        twf.load(tw_own + C::TW_IDX + lane);
        twi.load(tw_oth + C::TW_IDX + (63 - lane));
    } else {
        twf.base = tw_own + C::TW_IDX + lane;
        twi.base = tw_oth + C::TW_IDX + (63 - lane);
    }
    const int quarter = f * 2 + h; // which E/4 slots of the polynomial this wave decomposes

    // key words of this wave: [i][row p][c][level][f][h][e/2][lane] as double2
    const size_t part = (size_t)(GS::N / 2);
    const size_t bsk_step = (size_t)K1 * K1 * L * 4 * part;
    const double2 *bsk_w = reinterpret_cast<const double2 *>(bsk) + ((size_t)p * K1 * L * 4 + f * 2 + h) * part + lane;
    const uint32_t half_m1 = (1u << (logB - 1)) - 1u;
    [[maybe_unused]] const uint32_t bmask = (1u << logB) - 1u;

    STAMP_DECL
    for (int i = 0; i < n; i++) {
        const int a = __builtin_amdgcn_readfirstlane((int)MS[i]);
        if (a == 0) continue; // uniform over the workgroup
        STAMP_BEGIN
        [[maybe_unused]] const double2 *bp_i = bsk_w + (size_t)i * bsk_step;
        auto key = [&](int col_lev, int u) { return (bp_i + (size_t)col_lev * 4 * part)[u * 64]; };
        // ---- (1) digits of this wave's quarter of polynomial p -------------------------------
        if constexpr (SRC1) {
            // L = 1: the quarter is Q/2 PAIRS (j, j + N/2); stage 1 of the full transform is done here, once per pair
            // and field (U + V psi^(N/2) for half 0, U - V psi^(N/2) for half 1), and the four results go straight
            // into the transform scratches of the four waves that transform them - no digit buffer, no conversions and
            // no stage-1 product on the consumers' side (they used to compute all sixteen, each half redundantly)
            using FO = std::conditional_t<std::is_same<F, F0>::value, F1, F0>;
            double *x_f0 = X + (size_t)wave_of(p, f, 0) * GS::XPAD + lane, *x_f1 = X + (size_t)wave_of(p, f, 1) * GS::XPAD + lane;
            double *x_o0 = X + (size_t)wave_of(p, 1 - f, 0) * GS::XPAD + lane, *x_o1 = X + (size_t)wave_of(p, 1 - f, 1) * GS::XPAD + lane;
            auto digit = [&](int j) {
                const int src = (j - a) & (2 * N - 1);
                uint64_t v = acc_p[src & (N - 1)];
                if (src >= N) v = 0ull - v;
                v -= acc_p[j];
                const uint32_t st = (uint32_t)((v + (1ull << (63 - logB))) >> (64 - logB));
                return (double)((int)((st + half_m1) & bmask) - (int)half_m1); // st <= B/2 stays, above it st - B
            };
#pragma unroll
            for (int u = 0; u < Q / 2; u++) {
                const int e = quarter * (Q / 2) + u;
                const double U = digit(G::jA(lane, e)), D1 = digit(G::jA(lane, e + EH));
                const double Vf = stage1_digit_product<F>(D1, w1), Vo = stage1_digit_product<FO>(D1, w1o);
                x_f0[e * 64] = U + Vf;
                x_f1[e * 64] = U - Vf;
                x_o0[e * 64] = U + Vo;
                x_o1[e * 64] = U - Vo;
            }
        } else
#pragma unroll
        for (int u = 0; u < Q; u++) {
            const int j = G::jA(lane, quarter * Q + u);
            const int src = (j - a) & (2 * N - 1);
            uint64_t v = acc_p[src & (N - 1)];
            if (src >= N) v = 0ull - v;
            v -= acc_p[j];
            if constexpr (L == 1) {
                const uint32_t st = (uint32_t)((v + (1ull << (63 - logB))) >> (64 - logB));
                // nothing is left above the digit, the carry decides between d and d - B (d > B/2)
                dig_p[j] = (int)st - (int)(((st + half_m1) >> logB) << logB);
            } else { // least significant level first, the one-addition carry rule of pbs64_body
                const int rep = logB * L;
                uint32_t st = (uint32_t)((v + (1ull << (63 - rep))) >> (64 - rep));
#pragma unroll
                for (int lev = L - 1; lev >= 0; lev--) {
                    const uint32_t next = (st + half_m1 + __builtin_amdgcn_ubfe(st, 2 * logB - 1, 1)) >> logB;
                    dig_p[lev * N + j] = (dig_t)((int)st - (int)(next << logB));
                    st = next;
                }
            }
        }
        // first key column of this wave's half (E/4 double2 per column)
        double2 kw[K1][HC];
#pragma unroll
        for (int u = 0; u < HC; u++) kw[0][u] = key(0 * L, u);
        STAMP(0) // quarter decomposition
        lds_block_sync(); // digits published
        STAMP(1) // barrier 1
        // ---- (2) stage 1 of the full transform, half transform, products ---------------------
        // the two waves of a SIMD (polynomials 0 and 1 of one field and half) run every phase together;
        // stepping the issue priority down block by block keeps them abreast, so that neither finishes
        // the phase alone (a lone wave cannot hide its LDS latencies)
        double mine[EH], other[EH];
#pragma unroll
        for (int lev = 0; lev < L; lev++) {
            if (lev > 0) {
#pragma unroll
                for (int u = 0; u < HC; u++) kw[0][u] = key(0 * L + lev, u);
            }
            if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(PH);
            double x[1][EH];
            if constexpr (SRC1) {
#pragma unroll
                for (int e = 0; e < EH; e++) x[0][e] = xb[e * 64 + lane];
            } else {
                const dig_t *dg = dig_p + lev * N + lane;
#pragma unroll
                for (int e = 0; e < EH; e++) {
                    const double U = (double)dg[64 * e], V = stage1_digit_product<F>((double)dg[64 * (e + EH)], w1);
                    x[0][e] = h ? U - V : U + V;
                }
            }
            // the second column is asked for between the transform's second transpose and its last block: behind both
            // LDS round trips (fetching it before the transform was measured 1 % slower,
            // profiles/r02/si_kernel_experiments.txt), with a block of arithmetic to cover the latency
            auto fetch_kw1 = [&]() {
#pragma unroll
                for (int u = 0; u < HC; u++) kw[1][u] = key(1 * L + lev, u);
            };
#if HELM_SI_KW1_EARLY
            half_forward<F, LOGN - 1, decltype(twf), PH>(x, xb, twf, lane, fetch_kw1);
#else
            half_forward<F, LOGN - 1, decltype(twf), PH>(x, xb, twf, lane);
#endif
            if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(0);
            if (lev == 0) {
                STAMP(2) // digits read, stage 1, half transform
            }
#if !HELM_SI_KW1_EARLY
            fetch_kw1();
#endif
#pragma unroll
            for (int c = 0; c < K1; c++) {
#pragma unroll
                for (int u = 0; u < HC; u++) {
                    const double t0 = mulmod<F>(x[0][2 * u], kw[c][u].x), t1 = mulmod<F>(x[0][2 * u + 1], kw[c][u].y);
                    if (c == p) {
                        mine[2 * u] = lev == 0 ? t0 : mine[2 * u] + t0;
                        mine[2 * u + 1] = lev == 0 ? t1 : mine[2 * u + 1] + t1;
                    } else {
                        other[2 * u] = lev == 0 ? t0 : other[2 * u] + t0;
                        other[2 * u + 1] = lev == 0 ? t1 : other[2 * u + 1] + t1;
                    }
                }
            }
        }
        // a product is below 1.5 p (mulmod: (0.5 + 0.75 |a| 2^-52) p with |a| <= 9.6 p): the 2 L of a column are
        // summed as they are and recentred once (L <= 2: below 6 p)
#pragma unroll
        for (int e = 0; e < EH; e++) xb[e * 64 + lane] = other[e];
        STAMP(3) // products
        lds_block_sync();
#pragma unroll
        for (int e = 0; e < EH; e++) mine[e] = reduce<F>(mine[e] + x_poly[e * 64 + lane]);
        lds_block_sync(); // hand-over read: scratch free again
        STAMP(4) // barrier 2, sum, barrier 3
        // ---- (3) half inverse, meet the other half, last stage -------------------------------
        if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(PH);
        half_inverse<F, LOGN - 1, decltype(twi), PH, !(HELM_SI_LAZY_INV && HELM_SI_FUSED_XCHG)>(mine, xb, twi, lane); // a_h[e * 64 + lane], centred
        if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(0);
        STAMP(5) // half inverse
#pragma unroll
        for (int e = 0; e < EH; e++)
            if (!(HELM_SI_PAIR_LIFT && HELM_SI_FUSED_XCHG) || e / (EH / 4) != f * 2 + h) xb[e * 64 + lane] = mine[e]; // own slots stay
        lds_block_sync();
#if HELM_SI_FUSED_XCHG && HELM_SI_PAIR_LIFT
        lift_pairs<C, F, h, true>(mine, x_half, x_field, X + (size_t)wave_of(p, 1 - f, 1 - h) * GS::XPAD, acc_p, f, w1, w1o,
                                  p0inv_mod_p1, lane);
        lds_block_sync(); // accumulator complete, scratch free again
#elif HELM_SI_FUSED_XCHG
        // ---- (3b, 4) one exchange: every wave publishes its half inverse; the wave that lifts a slot computes the
        //      last inverse stage of that slot in BOTH fields (its own from registers + the other half, the other
        //      field's from the two waves that hold it) and the CRT.  Same values as the two-exchange form.
        {
            using FO = std::conditional_t<std::is_same<F, F0>::value, F1, F0>;
            const double *x_fh = X + (size_t)wave_of(p, 1 - f, 1 - h) * GS::XPAD; // other field, other half
            constexpr int HH = EH / 2;
#pragma unroll
            for (int e = 0; e < HH; e++) {
                const int s = f * HH + e;
                const double a = mine[s], o = x_half[s * 64 + lane];
                const double own = h ? reduce<F>(mulmod<F>(a - o, w1)) : reduce<F>(a + o);
                const double a2 = x_field[s * 64 + lane], o2 = x_fh[s * 64 + lane];
                const double oth = h ? reduce<FO>(mulmod<FO>(a2 - o2, w1o)) : reduce<FO>(a2 + o2);
                const double r0 = f == 0 ? own : oth, r1 = f == 0 ? oth : own;
                const double t = mulmod<F1>(r1 - r0, p0inv_mod_p1);
                const uint64_t xv = (uint64_t)to_int64(r0) + F0::P_U64 * (uint64_t)to_int64(t);
                acc_p[h * (N / 2) + s * 64 + lane] += xv;
            }
        }
        lds_block_sync(); // accumulator complete, scratch free again
#else
#pragma unroll
        for (int e = 0; e < EH; e++) {
            const double o = x_half[e * 64 + lane];
            // y[j] = a0 + a1 ; y[j + N/2] = (a0 - a1) * psi^-(N/2) = (a1 - a0) * psi^(N/2)
            mine[e] = h ? reduce<F>(mulmod<F>(mine[e] - o, w1)) : reduce<F>(mine[e] + o);
        }
        lds_block_sync(); // the other half has read: scratch free again
        STAMP(6) // half exchange, last stage, barriers 4 and 5
        // ---- (4) CRT: field-f wave lifts slots [f*EH/2, (f+1)*EH/2) of its half ----------------
        constexpr int HH = EH / 2;
#pragma unroll
        for (int e = 0; e < HH; e++) xb[e * 64 + lane] = mine[(1 - f) * HH + e];
        lds_block_sync();
#pragma unroll
        for (int e = 0; e < HH; e++) {
            const double own = mine[f * HH + e], oth = x_field[e * 64 + lane];
            const double r0 = f == 0 ? own : oth, r1 = f == 0 ? oth : own;
            const double t = mulmod<F1>(r1 - r0, p0inv_mod_p1);
            const uint64_t xv = (uint64_t)to_int64(r0) + F0::P_U64 * (uint64_t)to_int64(t);
            acc_p[h * (N / 2) + (f * HH + e) * 64 + lane] += xv;
        }
        lds_block_sync(); // accumulator complete
#endif
        STAMP(7) // CRT, barriers 6 and 7
    }
    STAMP_END((p * 2 + f) * 2 + h)
}

__host__ __device__ constexpr int bitrev_c(int v, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; b++) r |= ((v >> b) & 1) << (bits - 1 - b);
    return r;
}

// Multi-bit blind rotation (tfhe MultiBitPBS, grouping factor g <= 3; reference src/bin/helm.rs:83 installs
// the g = 3 set for arithmetic mode): per group of g mask words ONE external product
//     acc <- ( sum_S X^(e_S) * GGSW_S ) (x) acc,      e_S = sum_{i in S} a~_i,
// evaluated as  sum_S ( NTT(digits(acc)) .* M(e_S) ) .* K_S  in the transform domain: multiplying by the monomial
// X^e is a pointwise product with M(e)[j] = psi^(expo[j] * e), expo[j] the (odd) exponent of the evaluation
// point of spectrum position j (probed once per context through the key-conversion kernel) and psi_pow the 2N
// powers of psi.  Same wave roles, transforms, hand-over and CRT as pbs64s_body; n/g steps instead of n, the
// accumulator is replaced instead of added to.
template <typename C, typename F, int h, int P = -1>
__device__ __forceinline__ void pbs64s_mb_body(unsigned char *smem, const double *__restrict__ bsk, int n, int logB,
                                               double p0inv_mod_p1, double w1, double w1o, int p, int f, int lane, int g,
                                               const uint16_t *__restrict__ expo, const double *__restrict__ psi_pow)
{
    constexpr int LOGN = C::LOGN, K1 = C::K1;
    using G = typename C::G;
    using GS = typename C::GS;
    constexpr int PH = !C::PRIO ? 0 : 3;
    constexpr int N = G::N, E = G::E, EH = GS::E, Q = E / 4, HC = EH / 2;
    double *X = reinterpret_cast<double *>(smem + C::X_OFF);
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    using dig_t = typename C::dig_t;
    dig_t *DIG = reinterpret_cast<dig_t *>(smem + C::DIG_OFF);
    const uint16_t *MS = reinterpret_cast<const uint16_t *>(smem + C::MS_OFF);
    auto wave_of = [](int pp, int ff, int hh) { return (pp * 2 + ff) * 2 + hh; };
    double *xb = X + (size_t)wave_of(p, f, h) * GS::XPAD;
    const double *x_poly = X + (size_t)wave_of(1 - p, f, h) * GS::XPAD;
    const double *x_half = X + (size_t)wave_of(p, f, 1 - h) * GS::XPAD;
    const double *x_field = X + (size_t)wave_of(p, 1 - f, h) * GS::XPAD;
    uint64_t *acc_p = ACC + (size_t)p * N;
    [[maybe_unused]] dig_t *dig_p = DIG + (size_t)p * N;
    const double *twt = reinterpret_cast<const double *>(smem + C::TW_OFF);
    const double *tw_own = twt + (size_t)(f * 2 + h) * C::TW_PART, *tw_oth = twt + (size_t)(f * 2 + (1 - h)) * C::TW_PART;
    TwHybrid<LOGN - 1, false> twf{tw_own, tw_own + C::TW_IDX + lane};
    TwHybrid<LOGN - 1, true> twi{tw_oth, tw_oth + C::TW_IDX + (63 - lane)};
    const int quarter = f * 2 + h;
    const size_t part = (size_t)(GS::N / 2);
    const size_t bsk_step = (size_t)K1 * K1 * 4 * part; // one GGSW
    [[maybe_unused]] const double2 *bsk_w = reinterpret_cast<const double2 *>(bsk) + ((size_t)p * K1 * 4 + f * 2 + h) * part + lane;
    const uint32_t half_m1 = (1u << (logB - 1)) - 1u;
    [[maybe_unused]] const uint32_t bmask = (1u << logB) - 1u;
    const int subsets = 1 << g;
    // exponent of this lane's first spectrum position (slot e = 0 of the key-word order)
    const int c_lane = (int)expo[((size_t)h * HC * 64 + lane) * 2];
    // omega^k, k < EH (omega = psi^(2N/EH)), held by lane k: read with v_readlane by a uniform index
    const double om_tab = psi_pow[(lane & (EH - 1)) * (2 * N / EH)];
    const int om_lo = __double2loint(om_tab), om_hi = __double2hiint(om_tab);
#if HELM_SI_MB_NESTED
    // key words through buffer loads: descriptor in scalar registers, scalar byte offset per (group, subset, column)
    const unsigned ggsw_bytes = (unsigned)(bsk_step * 16), col_bytes = (unsigned)(4 * part * 16);
    const unsigned wave_off = (unsigned)(((size_t)p * K1 * 4 + f * 2 + h) * part * 16);
    KeyBuf kbuf;
    kbuf.init(bsk, (size_t)(n / g) * subsets * ggsw_bytes, lane);
#endif
    for (int t = 0; t < n / g; t++) {
        int am[3];
#pragma unroll
        for (int q = 0; q < 3; q++) am[q] = q < g ? __builtin_amdgcn_readfirstlane((int)MS[t * g + q]) : 0;
        // psi^(c_lane a_q) of the group's members: in flight during the decomposition and the transform
        double bq[3];
#pragma unroll
        for (int q = 0; q < 3; q++) bq[q] = psi_pow[(c_lane * am[q]) & (2 * N - 1)];
        // ---- (1) digits of this wave's quarter of polynomial p ----------------------------------
        // ---- (1) digits of this wave's quarter of polynomial p ----------------------------------
#pragma unroll
        for (int u = 0; u < Q; u++) {
            const int j = G::jA(lane, quarter * Q + u);
            const uint64_t v = acc_p[j];
            const uint32_t st = (uint32_t)((v + (1ull << (63 - logB))) >> (64 - logB));
            dig_p[j] = (int)st - (int)(((st + half_m1) >> logB) << logB);
        }
        lds_block_sync(); // digits published
        // ---- (2) stage 1 of the full transform, half transform ----------------------------------
        if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(PH);
        double x[1][EH];
        {
            const dig_t *dg = dig_p + lane;
#pragma unroll
            for (int e = 0; e < EH; e++) {
                const double U = (double)dg[64 * e], V = stage1_digit_product<F>((double)dg[64 * (e + EH)], w1);
                x[0][e] = h ? U - V : U + V;
            }
        }
        half_forward<F, LOGN - 1, decltype(twf), PH>(x, xb, twf, lane);
        if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(0);
#if HELM_SI_MB_NESTED
        // ---- the group's key in the transform domain, G[c] = sum_S M(e_S) .* K_S[p][c], in NESTED form, and the products
        //      x .* G[c].  M(e_S) is the product of its members' monomial vectors M_i = M(a~_i), so
        //        G = (K_000 + M_0 K_001) + M_1 (K_010 + M_0 K_011) + M_2 [(K_100 + M_0 K_101) + M_1 (K_110 + M_0 K_111)]:
        //      2^g - 1 modular multiplications per spectrum position and column instead of 2 (2^g - 1) (one per subset for
        //      the monomial's own value, one with the key word), plus g per position for the M_i.  The spectrum positions
        //      of a lane are expo(lane, e) = c_lane + (2N/EH) rev(e), so M_i[e] = psi^(c_lane a~_i) * omega^(a~_i rev(e)):
        //      one gathered power per lane and member (bq, fetched at the top of the step) times a wave-uniform power of
        //      omega read out of a lane-held table with v_readlane.  Walked slot pair by slot pair and column by column:
        //      the 2^g key words of one (slot pair, column) are fetched one iteration ahead (buffer loads: scalar offsets).
        static_assert(K1 == 2, "two key columns");
        double mine[EH], oth[EH];
        auto key_products = [&](auto g_const) {
            constexpr int GG = decltype(g_const)::value, SUB = 1 << GG; // compile-time group size: no branches in the walk
            const unsigned so_t = (unsigned)t * (unsigned)SUB * ggsw_bytes + wave_off;
            double2 kq[2][SUB];
            auto fetch = [&](int it, double2 (&dst)[SUB]) { // it = 2 u + c
                const int u = it >> 1, c = it & 1;
#pragma unroll
                for (int S = 0; S < SUB; S++)
                    dst[S] = kbuf.load(so_t + (unsigned)S * ggsw_bytes + (unsigned)c * col_bytes + (unsigned)(u >> 2) * 4096u, (u & 3) * 1024);
            };
            fetch(0, kq[0]);
            double m[GG][2];
#pragma unroll
            for (int it = 0; it < 2 * HC; it++) {
                const int u = it >> 1, c = it & 1;
                if (it + 1 < 2 * HC) fetch(it + 1, kq[(it + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (c == 0) { // the members' monomial values at the two positions of this slot pair
#pragma unroll
                    for (int q = 0; q < GG; q++)
#pragma unroll
                        for (int z = 0; z < 2; z++) {
                            const int k = (am[q] * bitrev_c(2 * u + z, GS::LOGE)) & (EH - 1);
                            const double om = __hiloint2double(__builtin_amdgcn_readlane(om_hi, k), __builtin_amdgcn_readlane(om_lo, k));
                            m[q][z] = mulmod<F>(bq[q], om);
                        }
                }
                // |K| <= 0.5 p, |M| <= 0.56 p: 1.1 p, 1.8 p, 2.4 p after the three levels - far below 2^53 = 14.2 p
                double2(&v)[SUB] = kq[it & 1];
#pragma unroll
                for (int q = 0; q < GG; q++)
#pragma unroll
                    for (int S = 0; S < SUB; S++)
                        if (!(S & ((2 << q) - 1))) {
                            v[S].x += mulmod<F>(v[S | (1 << q)].x, m[q][0]);
                            v[S].y += mulmod<F>(v[S | (1 << q)].y, m[q][1]);
                        }
                const double c0 = mulmod<F>(x[0][2 * u], reduce<F>(v[0].x)), c1 = mulmod<F>(x[0][2 * u + 1], reduce<F>(v[0].y)); // <= 1.5 p
                if (c == p) {
                    mine[2 * u] = c0;
                    mine[2 * u + 1] = c1;
                } else {
                    oth[2 * u] = c0;
                    oth[2 * u + 1] = c1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (g == 3) key_products(std::integral_constant<int, 3>());
        else key_products(std::integral_constant<int, 2>());
#pragma unroll
        for (int e = 0; e < EH; e++) xb[e * 64 + lane] = oth[e];
#else
        // ---- the group's key in the transform domain: G[c] = sum_S M(e_S) .* K_S[p][c], then the products
        //      x .* G[c].  The spectrum positions of a lane are expo(lane, e) = c_lane + (2N/EH) rev(e)
        //      (verified when the table is probed), so M(e_S)[e] = psi^(c_lane e_S) * omega^(e_S rev(e)) with
        //      omega = psi^(2N/EH) of order EH: one gathered power per lane and subset, EH wave-uniform ones.
        // The transformed digits wait in the wave's scratch meanwhile (their registers carry the key
        // pipeline: the two key polynomials of a subset are fetched one half-step ahead of their use).
        static_assert(K1 == 2, "two key columns, pipelined alternately");
#pragma unroll
        for (int e = 0; e < EH; e++) xb[e * 64 + lane] = x[0][e];
        double gcol[K1][EH];
        double2 ka[HC], kb[HC];
        {
            const double2 *kp = bsk_w + ((size_t)t * subsets) * bsk_step;
#pragma unroll
            for (int u = 0; u < HC; u++) ka[u] = kp[u * 64];
#pragma unroll
            for (int u = 0; u < HC; u++) kb[u] = (kp + (size_t)4 * part)[u * 64];
        }
        for (int S = 0; S < subsets; S++) {
            int e_s = 0;
#pragma unroll
            for (int q = 0; q < 3; q++)
                if ((S >> q) & 1) e_s += am[q];
            e_s &= 2 * N - 1;
            // the next subset's keys (the last iteration refetches its own: harmless, keeps the loop uniform)
            const double2 *kn = bsk_w + ((size_t)t * subsets + (S + 1 < subsets ? S + 1 : S)) * bsk_step;
            double mf[EH];
            if (S != 0) {
                // psi^(c_lane e_S) as the product of the members' powers (fetched at the top of the step);
                // omega^(e_S rev(e)) out of the lane-held table by wave-uniform index: no memory access here
                double bs = (S & 1) ? bq[0] : ((S & 2) ? bq[1] : bq[2]);
                if ((S & 1) && (S & 2)) bs = reduce<F>(mulmod<F>(bs, bq[1]));
                if ((S & 3) && (S & 4)) bs = reduce<F>(mulmod<F>(bs, bq[2]));
                // |m| <= 0.51 p, every term <= 0.51 p: G <= 4.1 p before the recentring below
#pragma unroll
                for (int e = 0; e < EH; e++) {
                    const int k = (e_s * bitrev_c(e, GS::LOGE)) & (EH - 1);
                    const double om = __hiloint2double(__builtin_amdgcn_readlane(om_hi, k), __builtin_amdgcn_readlane(om_lo, k));
                    mf[e] = mulmod<F>(bs, om);
                }
            }
#pragma unroll
            for (int u = 0; u < HC; u++) {
                gcol[0][2 * u] = S == 0 ? ka[u].x : gcol[0][2 * u] + mulmod<F>(ka[u].x, mf[2 * u]);
                gcol[0][2 * u + 1] = S == 0 ? ka[u].y : gcol[0][2 * u + 1] + mulmod<F>(ka[u].y, mf[2 * u + 1]);
            }
#pragma unroll
            for (int u = 0; u < HC; u++) ka[u] = kn[u * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < HC; u++) {
                gcol[1][2 * u] = S == 0 ? kb[u].x : gcol[1][2 * u] + mulmod<F>(kb[u].x, mf[2 * u]);
                gcol[1][2 * u + 1] = S == 0 ? kb[u].y : gcol[1][2 * u + 1] + mulmod<F>(kb[u].y, mf[2 * u + 1]);
            }
#pragma unroll
            for (int u = 0; u < HC; u++) kb[u] = (kn + (size_t)4 * part)[u * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_wave_sync();
#pragma unroll
        for (int e = 0; e < EH; e++) x[0][e] = xb[e * 64 + lane];
        lds_wave_sync();
        double col[K1][EH];
#pragma unroll
        for (int c = 0; c < K1; c++)
#pragma unroll
            for (int e = 0; e < EH; e++) col[c][e] = mulmod<F>(x[0][e], reduce<F>(gcol[c][e])); // <= 1.5 p
        double mine[EH];
#pragma unroll
        for (int e = 0; e < EH; e++) {
            xb[e * 64 + lane] = p == 0 ? col[1][e] : col[0][e];
            mine[e] = p == 0 ? col[0][e] : col[1][e];
        }
#endif
        lds_block_sync();
#pragma unroll
        for (int e = 0; e < EH; e++) mine[e] = reduce<F>(mine[e] + x_poly[e * 64 + lane]); // <= 11.4 p before
        lds_block_sync(); // hand-over read: scratch free again
        // ---- (3) half inverse, meet the other half, last stage -----------------------------------
        if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(PH);
        half_inverse<F, LOGN - 1, decltype(twi), PH, !(HELM_SI_LAZY_INV && HELM_SI_FUSED_XCHG)>(mine, xb, twi, lane);
        if constexpr (C::PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int e = 0; e < EH; e++)
            if (!(HELM_SI_PAIR_LIFT && HELM_SI_FUSED_XCHG) || e / (EH / 4) != f * 2 + h) xb[e * 64 + lane] = mine[e]; // own slots stay
        lds_block_sync();
#if HELM_SI_FUSED_XCHG
#if HELM_SI_PAIR_LIFT
        lift_pairs<C, F, h, false>(mine, x_half, x_field, X + (size_t)wave_of(p, 1 - f, 1 - h) * GS::XPAD, acc_p, f, w1, w1o,
                                   p0inv_mod_p1, lane);
#else
        // ---- (3b, 4) one exchange: every wave publishes its half inverse; the wave that lifts a slot computes the
        //      last inverse stage of that slot in BOTH fields (its own from registers + the other half, the other
        //      field's from the two waves that hold it) and the CRT.  Same values as the two-exchange form.
        {
            using FO = std::conditional_t<std::is_same<F, F0>::value, F1, F0>;
            const double *x_fh = X + (size_t)wave_of(p, 1 - f, 1 - h) * GS::XPAD; // other field, other half
            constexpr int HH = EH / 2;
#pragma unroll
            for (int e = 0; e < HH; e++) {
                const int s = f * HH + e;
                const double a = mine[s], o = x_half[s * 64 + lane];
                const double own = h ? reduce<F>(mulmod<F>(a - o, w1)) : reduce<F>(a + o);
                const double a2 = x_field[s * 64 + lane], o2 = x_fh[s * 64 + lane];
                const double oth = h ? reduce<FO>(mulmod<FO>(a2 - o2, w1o)) : reduce<FO>(a2 + o2);
                const double r0 = f == 0 ? own : oth, r1 = f == 0 ? oth : own;
                const double t = mulmod<F1>(r1 - r0, p0inv_mod_p1);
                const uint64_t xv = (uint64_t)to_int64(r0) + F0::P_U64 * (uint64_t)to_int64(t);
                acc_p[h * (N / 2) + s * 64 + lane] = xv;
            }
        }
#endif
        lds_block_sync(); // accumulator complete, scratch free again
#else
#pragma unroll
        for (int e = 0; e < EH; e++) {
            const double o = x_half[e * 64 + lane];
            mine[e] = h ? reduce<F>(mulmod<F>(mine[e] - o, w1)) : reduce<F>(mine[e] + o);
        }
        lds_block_sync();
        // ---- (4) CRT: the lifted value REPLACES the accumulator ----------------------------------
        constexpr int HH = EH / 2;
#pragma unroll
        for (int e = 0; e < HH; e++) xb[e * 64 + lane] = mine[(1 - f) * HH + e];
        lds_block_sync();
#pragma unroll
        for (int e = 0; e < HH; e++) {
            const double own = mine[f * HH + e], oth = x_field[e * 64 + lane];
            const double r0 = f == 0 ? own : oth, r1 = f == 0 ? oth : own;
            const double tt = mulmod<F1>(r1 - r0, p0inv_mod_p1);
            acc_p[h * (N / 2) + (f * HH + e) * 64 + lane] = (uint64_t)to_int64(r0) + F0::P_U64 * (uint64_t)to_int64(tt);
        }
        lds_block_sync(); // accumulator complete
#endif
    }
}

template <typename C, bool MB>
__global__ __launch_bounds__(64 * C::NW, 1) void k_pbs64s(const Pbs64Job *__restrict__ jobs,
                                                          const uint64_t *__restrict__ small,
                                                          const uint64_t *__restrict__ luts,
                                                          const double *__restrict__ bsk,    // split layout
                                                          const double *__restrict__ tw_sub, // [2 fields][2 halves][N/2]
                                                          const double *__restrict__ tw0, const double *__restrict__ tw1,
                                                          uint64_t *__restrict__ out, int n, int logB,
                                                          double p0inv_mod_p1, int g,
                                                          const uint16_t *__restrict__ expo, // multi-bit: [2 halves][N/2]
                                                          const double *__restrict__ psi_pow) // multi-bit: [2 fields][2N]
{
    constexpr int LOGN = C::LOGN, K = C::K;
    using G = typename C::G;
    using GS = typename C::GS;
    constexpr int N = G::N, E = G::E;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *ACC = reinterpret_cast<uint64_t *>(smem + C::ACC_OFF);
    uint16_t *MS = reinterpret_cast<uint16_t *>(smem + C::MS_OFF);
    double *TW = reinterpret_cast<double *>(smem + C::TW_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the waves of polynomial 1 take the halves the other way round: each SIMD (wave index mod 4) then holds one wave of
    // either half - the last inverse stage costs the h = 1 waves sixteen modular multiplications more than the h = 0 waves
    const int p = w >> 2, f = (w >> 1) & 1, h = HELM_SI_MIX_HALVES ? ((w & 1) ^ p) : (w & 1);
    const Pbs64Job job = jobs[blockIdx.x];
    const uint64_t *lwe = small + (size_t)job.in_row * ((size_t)n + 1);
    for (int i = tid; i <= n; i += 64 * C::NW) MS[i] = (uint16_t)modswitch64(lwe[i], LOGN + 1);
    if (tid < 2 * C::NW) reinterpret_cast<uint32_t *>(smem + C::FLAG_OFF)[tid] = 0u;
    // derived tables: index part for blocks A, B and lane table for block C, per (field, half)
    for (int q = 0; q < 4; q++) {
        const double *src = tw_sub + (size_t)q * GS::N;
        double *dst = TW + (size_t)q * C::TW_PART;
        for (int i = tid; i < C::TW_IDX; i += 64 * C::NW) dst[i] = src[i];
        for (int r = w; r < GS::TWC; r += C::NW) dst[C::TW_IDX + r * 64 + lane] = src[tw_lane_index<LOGN - 1>(GS::TWB + r, lane)];
    }
    __syncthreads();
    {
        const int bt = (int)MS[n];
        const uint64_t *tv = luts + (size_t)job.lut * N;
        for (int j = tid; j < (K + 1) * N; j += 64 * C::NW) {
            uint64_t v = 0;
            if (j >= K * N) {
                const int idx = ((j - K * N) + bt) & (2 * N - 1);
                v = tv[idx & (N - 1)];
                if (idx >= N) v = 0ull - v;
            }
            ACC[j] = v;
        }
    }
    __syncthreads();
    // psi^(N/2): entry 1 of the full forward table of this wave's field
    const double w1 = f == 0 ? tw0[1] : tw1[1], w1o = f == 0 ? tw1[1] : tw0[1];
    // one specialisation per (field, transform half): both are uniform over the wave
    // (the polynomial as a literal too where HELM_SI_STATIC_P: the bodies are inlined, so `c == p` in the products and the
    // row offsets fold - 66 v_cndmask per wave-step gone from the classical kernel)
#define HELM_SI_BODY(FN, FT, HH, FF, ...)                                                      \
    do {                                                                                       \
        if (HELM_SI_STATIC_P && p == 0) FN<C, FT, HH, 0>(smem, bsk, n, logB, p0inv_mod_p1, w1, w1o, 0, FF, lane, ##__VA_ARGS__); \
        else if (HELM_SI_STATIC_P) FN<C, FT, HH, 1>(smem, bsk, n, logB, p0inv_mod_p1, w1, w1o, 1, FF, lane, ##__VA_ARGS__);      \
        else FN<C, FT, HH>(smem, bsk, n, logB, p0inv_mod_p1, w1, w1o, p, FF, lane, ##__VA_ARGS__);                            \
    } while (0)
    if constexpr (MB) {
        if (f == 0) {
            if (h == 0) HELM_SI_BODY(pbs64s_mb_body, F0, 0, 0, g, expo, psi_pow);
            else HELM_SI_BODY(pbs64s_mb_body, F0, 1, 0, g, expo, psi_pow);
        } else {
            if (h == 0) HELM_SI_BODY(pbs64s_mb_body, F1, 0, 1, g, expo, psi_pow + 2 * N);
            else HELM_SI_BODY(pbs64s_mb_body, F1, 1, 1, g, expo, psi_pow + 2 * N);
        }
    } else if (f == 0) {
        if (h == 0) HELM_SI_BODY(pbs64s_body, F0, 0, 0);
        else HELM_SI_BODY(pbs64s_body, F0, 1, 0);
    } else {
        if (h == 0) HELM_SI_BODY(pbs64s_body, F1, 0, 1);
        else HELM_SI_BODY(pbs64s_body, F1, 1, 1);
    }
#undef HELM_SI_BODY

    uint64_t *ob = out + (size_t)job.out_row * ((size_t)K * N + 1);
    const uint64_t *acc_p = ACC + (size_t)p * N;
    const int quarter = f * 2 + h;
    if (p < K) {
#pragma unroll
        for (int u = 0; u < E / 4; u++) {
            const int j = G::jA(lane, quarter * (E / 4) + u);
            const uint64_t v = acc_p[j];
            if (j == 0) ob[p * N] = v;
            else ob[p * N + (N - j)] = 0ull - v;
        }
    } else if (w == K * 4 && lane == 0) {
        ob[K * N] = acc_p[0];
    }
}

// key conversion for k_pbs64s: standard-domain u64 -> field F, stage 1 + half transform h, times N^-1:
//   dst[i][r][c][lev][f][h][e/2][lane][e&1]    (src is [i][lev][r][c][N]); one wave per (polynomial, half)
template <typename F, int LOGN>
__global__ __launch_bounds__(64) void k_bsk_convert64s(const uint64_t *__restrict__ src, double *__restrict__ dst,
                                                       const double *__restrict__ tw_full, const double *__restrict__ tw_sub_f,
                                                       double n_inv, double two32, int K1, int f, int L)
{
    using G = Geo<LOGN>;
    using GS = Geo<LOGN - 1>;
    constexpr int N = G::N, EH = GS::E;
    __shared__ double xbuf[GS::XPAD];
    const int lane = threadIdx.x;
    const size_t poly = blockIdx.x >> 1;
    const int h = blockIdx.x & 1;
    const int c = poly % K1;
    const int r = (poly / K1) % K1;
    const int lev = (poly / ((size_t)K1 * K1)) % L;
    const size_t i = poly / ((size_t)K1 * K1 * L);
    const double w1 = tw_full[1];
    auto load = [&](int j) {
        const uint64_t v = src[poly * N + j];
        const double hi = (double)(int32_t)(uint32_t)(v >> 32), lo = (double)(uint32_t)v;
        return reduce<F>(mulmod<F>(hi, two32) + lo);
    };
    double x[1][EH];
#pragma unroll
    for (int e = 0; e < EH; e++) {
        const double U = load(e * 64 + lane), V = mulmod<F>(load(e * 64 + lane + N / 2), w1);
        x[0][e] = reduce<F>(h ? U - V : U + V);
    }
    ntt_forward<F, LOGN - 1, 1>(x, xbuf, TwMem{tw_sub_f + (size_t)h * GS::N}, lane);
    const size_t dpoly = (((((i * K1 + r) * K1 + c) * L + lev) * 2 + f) * 2 + h);
    double *d = dst + dpoly * GS::N;
#pragma unroll
    for (int e = 0; e < EH; e++) d[((e >> 1) * 64 + lane) * 2 + (e & 1)] = reduce<F>(mulmod<F>(x[0][e], n_inv));
}

// ------------------------------------------------------------------------------------
// k_keyswitch64: as k_keyswitch of helm_hip.hip, 64-bit words.  Grid (ceil(jobs/4),
// column chunks of 256); digits of four ciphertexts packed as 4 x int8 per LDS word.
//   out[c] = (c == n ? body : 0) - sum_t sum_j digit(t,j) * KSK[t][j][c]
// ------------------------------------------------------------------------------------
template <int KSL>
__global__ __launch_bounds__(256) void k_keyswitch64(const Ks64Job *__restrict__ jobs, const uint64_t *__restrict__ big,
                                                     const uint64_t *__restrict__ ksk, uint64_t *__restrict__ out,
                                                     int n, int kN, int logB, int count, int t_chunk)
{
    // blockIdx.z: slice of t_chunk input coefficients (see k_keyswitch of helm_hip.hip); with more
    // than one slice the partial sums meet in `out` (zeroed beforehand) by 64-bit atomic add
    constexpr int GN = 4;
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *DIG = reinterpret_cast<uint32_t *>(smem); // [t_chunk * KSL]
    const int g0 = blockIdx.x * GN;
    const int ng = min(GN, count - g0);
    const int t0 = blockIdx.z * t_chunk, t1 = min(kN, t0 + t_chunk);
    Ks64Job job[GN];
#pragma unroll
    for (int g = 0; g < GN; g++) job[g] = jobs[g0 + (g < ng ? g : 0)];
    const size_t brow = (size_t)kN + 1;
    const int rep = logB * KSL;
    const uint64_t mask = (1ull << logB) - 1ull;
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        uint32_t packed[KSL];
#pragma unroll
        for (int j = 0; j < KSL; j++) packed[j] = 0;
#pragma unroll
        for (int g = 0; g < GN; g++) {
            if (g < ng) {
                const uint64_t v = big[brow * (size_t)job[g].in_row + t];
                uint64_t state = (v + (1ull << (63 - rep))) >> (64 - rep);
#pragma unroll
                for (int lev = KSL - 1; lev >= 0; lev--) {
                    const uint64_t d = state & mask;
                    state >>= logB;
                    const uint64_t carry = (((d - 1ull) | state) & d) >> (logB - 1);
                    state += carry;
                    const int dig = (int)(uint32_t)d - (int)((uint32_t)carry << logB);
                    packed[lev] |= ((uint32_t)dig & 0xFFu) << (8 * g);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KSL; j++) DIG[(t - t0) * KSL + j] = packed[j];
    }
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c > n) return;
    const size_t krow = (size_t)n + 1;
    uint64_t acc[GN] = {0, 0, 0, 0};
    const uint64_t *kp = ksk + (size_t)t0 * KSL * krow + c;
    const int rows = (t1 - t0) * KSL;
#pragma unroll 4
    for (int r = 0; r < rows; r++) {
        const uint64_t kw = kp[(size_t)r * krow];
        const uint32_t pk = DIG[r];
#pragma unroll
        for (int g = 0; g < GN; g++) acc[g] += (uint64_t)(int64_t)(int32_t)__builtin_amdgcn_sbfe(pk, 8 * g, 8) * kw;
    }
#pragma unroll
    for (int g = 0; g < GN; g++) {
        if (g < ng) {
            const uint64_t body = (c == n && blockIdx.z == 0) ? big[brow * (size_t)job[g].in_row + kN] : 0ull;
            uint64_t *dst = out + krow * (size_t)job[g].out_row + c;
            if (gridDim.z == 1) *dst = body - acc[g];
            else atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)(body - acc[g]));
        }
    }
}

// ------------------------------------------------------------------------------------
// Keyswitch on the matrix cores (as k_ks_digits / k_ks_mfma of helm_hip.hip, 64-bit words): the key's EIGHT
// byte planes as signed bytes (K_b - 128), the level count padded to LP in {1, 2, 4, 8} (a lane's 16 fragment
// bytes hold whole words' digits; padded levels carry zero digits), int32 accumulators per plane, exact:
//   sum_r d K = sum_b 2^(8b) sum_r d (K_b - 128) + 0x8080808080808080 * sum_r d      (mod 2^64)
// One wave = 64 ciphertexts x 16 key columns x 8 planes (128 accumulator registers).
// ------------------------------------------------------------------------------------
typedef int v4i64k __attribute__((ext_vector_type(4)));

template <int KSL>
__global__ __launch_bounds__(256) void k_ks64_digits(const Ks64Job *__restrict__ jobs, const uint64_t *__restrict__ big,
                                                     int8_t *__restrict__ dig, int32_t *__restrict__ dsum,
                                                     uint64_t *__restrict__ body, int kN, int logB, int count, int kchunks)
{
    constexpr int LP = KSL <= 1 ? 1 : KSL <= 2 ? 2 : KSL <= 4 ? 4 : 8;
    const int g = blockIdx.x; // padded index; g >= count: an all-zero row
    __shared__ int red[256];
    int local = 0;
    const size_t brow = (size_t)kN + 1;
    const bool live = g < count;
    Ks64Job job{};
    if (live) job = jobs[g];
    int8_t *tile = dig + (size_t)(g >> 4) * kchunks * 1024;
    const int rep = logB * KSL;
    const uint64_t mask = (1ull << logB) - 1ull;
    for (int t = threadIdx.x; t < kN; t += 256) {
        int d[LP];
#pragma unroll
        for (int j = 0; j < LP; j++) d[j] = 0;
        if (live) {
            const uint64_t v = big[brow * (size_t)job.in_row + t];
            uint64_t state = (v + (1ull << (63 - rep))) >> (64 - rep);
#pragma unroll
            for (int lev = KSL - 1; lev >= 0; lev--) {
                const uint64_t dd = state & mask;
                state >>= logB;
                const uint64_t carry = (((dd - 1ull) | state) & dd) >> (logB - 1);
                state += carry;
                d[lev] = (int)(uint32_t)dd - (int)((uint32_t)carry << logB);
            }
        }
        const int r0 = t * LP, kc = r0 >> 6, kq = (r0 & 63) >> 4, j0 = r0 & 15;
        int8_t *dst = tile + ((size_t)kc * 64 + (kq << 4 | (g & 15))) * 16 + j0;
#pragma unroll
        for (int j = 0; j < LP; j++) {
            dst[j] = (int8_t)d[j];
            local += d[j];
        }
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dsum[g] = red[0];
        body[g] = live ? big[brow * (size_t)job.in_row + kN] : 0ull;
    }
}

__global__ __launch_bounds__(64) void k_ks64_mfma(const Ks64Job *__restrict__ jobs, const int8_t *__restrict__ dig,
                                                  const int32_t *__restrict__ dsum, const uint64_t *__restrict__ body,
                                                  const int8_t *__restrict__ kplanes, uint64_t *__restrict__ out, int n,
                                                  int count, int kchunks, int ctiles)
{
    constexpr int GT = 4; // gate tiles of 16 per wave; one column tile of 16; eight byte planes
    const int lane = threadIdx.x;
    const int gt0 = blockIdx.x * GT, ct = blockIdx.y;
    const v4i64k *A = reinterpret_cast<const v4i64k *>(dig) + (size_t)gt0 * kchunks * 64 + lane;
    const v4i64k *B = reinterpret_cast<const v4i64k *>(kplanes) + (size_t)ct * kchunks * 64 + lane;
    const size_t plane = (size_t)ctiles * kchunks * 64;
    v4i64k acc[GT][8];
#pragma unroll
    for (int a = 0; a < GT; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = v4i64k{0, 0, 0, 0};
    for (int kc = 0; kc < kchunks; kc++) {
        v4i64k fa[GT], fb[8];
#pragma unroll
        for (int a = 0; a < GT; a++) fa[a] = A[((size_t)a * kchunks + kc) * 64];
#pragma unroll
        for (int b = 0; b < 8; b++) fb[b] = B[(size_t)b * plane + (size_t)kc * 64];
#pragma unroll
        for (int a = 0; a < GT; a++)
#pragma unroll
            for (int b = 0; b < 8; b++) acc[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    const size_t krow = (size_t)n + 1;
    const int col = ct * 16 + (lane & 15);
#pragma unroll
    for (int a = 0; a < GT; a++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int g = (gt0 + a) * 16 + (lane >> 4) * 4 + reg; // C/D map: column = lane & 15, row = (lane >> 4) * 4 + reg
            if (g >= count || col > n) continue;
            uint64_t s = 0x8080808080808080ull * (uint64_t)(int64_t)dsum[g];
#pragma unroll
            for (int b = 0; b < 8; b++) s += (uint64_t)(int64_t)acc[a][b][reg] << (8 * b);
            out[krow * (size_t)jobs[g].out_row + col] = (col == n ? body[g] : 0ull) - s;
        }
}

// out row = sum coef * in rows + const (body only).  One workgroup per output row.
__global__ __launch_bounds__(256) void k_lincomb64(const int32_t *__restrict__ in_idx, const int64_t *__restrict__ coef,
                                                   const uint64_t *__restrict__ body_add,
                                                   const int32_t *__restrict__ out_idx, const uint64_t *__restrict__ src,
                                                   uint64_t *__restrict__ dst, int terms, int dim)
{
    const int g = blockIdx.x;
    const size_t row = (size_t)dim + 1;
    uint64_t *o = dst + row * (size_t)(out_idx ? out_idx[g] : g);
    for (int i = threadIdx.x; i <= dim; i += 256) {
        uint64_t v = i == dim ? body_add[g] : 0ull;
        for (int t = 0; t < terms; t++) {
            const int r = in_idx[(size_t)g * terms + t];
            if (r >= 0) v += (uint64_t)coef[(size_t)g * terms + t] * src[row * (size_t)r + i];
        }
        o[i] = v;
    }
}

// a gate of a level must not read a row another gate of the level writes; in-place rows
// (out == in of the SAME gate) are common in the radix layer, so sums go through a staging
// buffer first when asked to
__global__ __launch_bounds__(256) void k_rows64(const uint64_t *__restrict__ src, const int32_t *__restrict__ src_row,
                                                uint64_t *__restrict__ dst, const int32_t *__restrict__ dst_row,
                                                int dim)
{
    const size_t row = (size_t)dim + 1;
    const int s = src_row ? src_row[blockIdx.x] : (int)blockIdx.x;
    const int d = dst_row ? dst_row[blockIdx.x] : (int)blockIdx.x;
    if (d < 0) return;
    for (int i = threadIdx.x; i <= dim; i += 256) dst[row * (size_t)d + i] = s < 0 ? 0ull : src[row * (size_t)s + i];
}

__global__ __launch_bounds__(256) void k_set_trivial64(const int32_t *__restrict__ idx, const uint64_t *__restrict__ body,
                                                       uint64_t *__restrict__ wires, int dim)
{
    const size_t row = (size_t)dim + 1;
    uint64_t *dst = wires + row * (size_t)idx[blockIdx.x];
    for (int i = threadIdx.x; i <= dim; i += 256) dst[i] = i == dim ? body[blockIdx.x] : 0ull;
}

// standard-domain u64 (taken as signed) -> NTT domain of field F, times N^-1, in the lane
// order k_pbs64 reads: dst[i][r][c][lev][f][e/2][lane][e&1]   (src is [i][lev][r][c][N])
template <typename F, int LOGN>
__global__ __launch_bounds__(64) void k_bsk_convert64(const uint64_t *__restrict__ src, double *__restrict__ dst,
                                                      const double *__restrict__ tw_fwd, double n_inv, double two32,
                                                      int K1, int L, int f)
{
    using G = Geo<LOGN>;
    constexpr int N = G::N, E = G::E;
    __shared__ double xbuf[G::XPAD];
    const int lane = threadIdx.x;
    const size_t poly = blockIdx.x;
    const int c = poly % K1;
    const int r = (poly / K1) % K1;
    const int lev = (poly / ((size_t)K1 * K1)) % L;
    const size_t i = poly / ((size_t)K1 * K1 * L);
    double x[1][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint64_t v = src[poly * N + G::jA(lane, e)];
        // v = hi * 2^32 + lo with hi signed: reduce in the field
        const double hi = (double)(int32_t)(uint32_t)(v >> 32), lo = (double)(uint32_t)v;
        x[0][e] = reduce<F>(mulmod<F>(hi, two32) + lo);
    }
    ntt_forward<F, LOGN, 1>(x, xbuf, TwMem{tw_fwd}, lane);
    const size_t dpoly = (((i * K1 + r) * K1 + c) * L + lev) * 2 + f;
    double *d = dst + dpoly * N;
#pragma unroll
    for (int e = 0; e < E; e++) d[((e >> 1) * 64 + lane) * 2 + (e & 1)] = reduce<F>(mulmod<F>(x[0][e], n_inv));
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
typedef unsigned __int128 u128;
uint64_t mulmod_u64(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((u128)a * b % p); }
uint64_t powmod_u64(uint64_t a, uint64_t e, uint64_t p)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulmod_u64(r, a, p);
        a = mulmod_u64(a, a, p);
        e >>= 1;
    }
    return r;
}
double centred(uint64_t v, uint64_t p) { return v > p / 2 ? (double)((int64_t)v - (int64_t)p) : (double)(int64_t)v; }
int bitrev(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max(n + n / 2, (size_t)64);
        if (hipMalloc(&p, want * sizeof(T)) != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace

struct helm_si_wires {
    helm_si_ctx *owner;
    uint64_t *d;
    int64_t n_rows;
};

struct helm_si_ctx {
    int device = 0;
    helm_si_ctx *lane_of = nullptr; // helm_si_ctx_fork(): the context whose keys and tables this one shares
    helm_si_params P{};
    int logN = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    bool high_priority = false; // own_stream was created at the device's highest priority (helm_si_set_priority)
    double *tw[2] = {nullptr, nullptr};
    double n_inv[2] = {0, 0}, two32[2] = {0, 0};
    double p0inv_mod_p1 = 0;
    int pair = 0;                // CRT pair of the tables above: 0 = FpG / FpG2 (49 bits), 1 = FpJ / FpJ2 (46 bits: k_pbs64k contexts
                                 // whose loaded key fits, helm_si_load_bootstrap_key)
    double *bsk = nullptr;
    double *bsk_split = nullptr; // layout of k_pbs64s (N >= 1024, pbs_l = 1)
    double *tw_sub = nullptr;    // derived half-transform tables [2 fields][2 halves][N/2]
    int group = 1;               // multi-bit grouping factor (1: classical blind rotation)
    uint16_t *expo = nullptr;    // multi-bit: exponent of the evaluation point per spectrum position [2 halves][N/2]
    double *psi_pow = nullptr;   // multi-bit: psi^t, t < 2N, per field
    bool use_split = false;
    uint64_t *ksk = nullptr;
    int8_t *ksk_planes = nullptr; // matrix-core keyswitch: eight byte planes as signed bytes, B-fragment order
    int ks_kchunks = 0, ks_ctiles = 0, ks_mfma = 1; // HELM_HIP_KS_MFMA=0: the vector-ALU keyswitch for every launch
    DevBuf<int8_t> d_ksdig;
    DevBuf<int32_t> d_ksdsum;
    DevBuf<uint64_t> d_ksbody;
    bool have_bsk = false, have_ksk = false;
    uint64_t delta = 0;
    int n_cus = 256;
    int64_t round_capacity = 0; // bootstraps resident at once (CUs x workgroups per CU of the set's kernel), probed on first need
    // per-call scratch
    DevBuf<Pbs64Job> d_pbs;
    DevBuf<Ks64Job> d_ks;
    DevBuf<uint64_t> d_small, d_luts, d_stage, d_body;
    DevBuf<int32_t> d_idx, d_idx2;
    DevBuf<int64_t> d_coef;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pbs, ev_ks, ev_lin;
    helm_si_timing tacc{};
    std::vector<helm_si_wires *> child_wires; // released with the context (see helm_hip.hip)
    // Call slots: helm_si_lincomb / helm_si_apply_luts copy their index arrays into a slot's pinned arena, issue ONE
    // asynchronous copy of it and launch - no stream synchronisation inside the call, so the host plans the next
    // batched round of the radix layer while the GPU runs this one (a round used to leave the GPU idle for ~1 ms of
    // host planning, copies and synchronisations out of ~8 ms).  A slot is reused three calls later, after its event.
    struct CallSlot {
        char *host = nullptr, *dev = nullptr;
        size_t cap = 0, used = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    } slots[3];
    int next_slot = 0;
    uint64_t luts_hash = 0; // of the look-up tables resident in d_luts (re-uploaded only when they change)
    std::vector<uint64_t> luts_host; // host copy of the resident tables: compared word for word when the hash matches
    size_t luts_words = 0;
    // multi-GPU (helm_si_set_exchange): every bootstrap batch of at least x_min ciphertexts is split
    // into x_world contiguous chunks, this rank bootstraps chunk x_rank into x_stage, the caller's
    // collective fills x_gather with every rank's chunk and the rows are scattered into the table
    // helm_si_set_audit: every helm_si_lincomb / helm_si_apply_luts call hands its operand rows and results to the host
    helm_si_audit_fn audit_fn = nullptr;
    void *audit_user = nullptr;
    int x_rank = 0, x_world = 1;
    int64_t x_min = 0, x_cap = 0;
    uint64_t *x_stage = nullptr, *x_gather = nullptr;
    helm_si_exchange_fn x_fn = nullptr;
    void *x_user = nullptr;
    // helm_si_set_exchange_comm: the collective is the library's own ncclAllGather through x_comm; the gather buffer is
    // the context's (x_own), this rank's chunk is computed straight into its slot of it (all-gather in place)
    helm_comm *x_comm = nullptr;
    uint64_t *x_own = nullptr;
    bool x_on = false; // sharding active (world = 1 with a collective installed counts: the single-GPU test of the path)
    int64_t x_batches = 0, x_rows = 0;
    // where this rank's chunk of a sharded batch goes, and the exchange that follows it
    uint64_t *x_slot(int64_t rows) const
    {
        return x_comm ? x_gather + (size_t)x_rank * (size_t)rows * ((size_t)P.k * P.N + 1) : x_stage;
    }
};

namespace {

// a lane works on the tables of the context it was forked from
bool owns(const helm_si_ctx *ctx, const helm_si_wires *w)
{
    return w->owner == ctx || (ctx->lane_of && w->owner == ctx->lane_of);
}

struct Timed {
    helm_si_ctx *ctx;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> *list;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(helm_si_ctx *c, std::vector<std::pair<hipEvent_t, hipEvent_t>> *l) : ctx(c), list(l)
    {
        if (ctx->timing) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~Timed()
    {
        if (ctx->timing) {
            (void)hipEventRecord(b, ctx->stream);
            list->push_back({a, b});
        }
    }
};

bool si_supported(const helm_si_params &P)
{
    // k > 1: k_pbs64k, one level, N = 512 (PARAM_MESSAGE_1_CARRY_1_KS_PBS of helm.rs:301 has k = 3)
    if (P.k == 2 || P.k == 3) return P.N == 512 && P.pbs_l == 1 && P.grouping_factor <= 1;
    if (P.k != 1) return false;
    if (!(P.N == 512 || P.N == 1024 || P.N == 2048)) return false;
    return P.pbs_l == 1 || P.pbs_l == 2;
}

template <typename C>
hipError_t launch_pbs64k_c(helm_si_ctx *ctx, const Pbs64Job *jobs, int64_t count, const uint64_t *small,
                           const uint64_t *luts, uint64_t *out, int *per_cu = nullptr)
{
    static std::atomic<bool> attr_done[64]; // (rank threads of one process launch concurrently)
    // the CRT pair follows the key loaded into the PRIMARY context (a lane shares its key and tables: helm_si_ctx_fork; a key
    // loaded after the fork may have moved the pair)
    const helm_si_ctx *root = ctx->lane_of ? ctx->lane_of : ctx;
    auto kern = root->pair ? k_pbs64k<C, J0, J1> : k_pbs64k<C, F0, F1>;
    if (!attr_done[ctx->device & 63]) {
        for (auto kk : {k_pbs64k<C, F0, F1>, k_pbs64k<C, J0, J1>}) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)C::BYTES);
            if (e != hipSuccess) return e;
        }
        attr_done[ctx->device & 63] = true;
    }
    if (per_cu) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, 64 * C::NW, C::BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(64 * C::NW), C::BYTES, ctx->stream, jobs, small, luts, root->bsk,
                       root->tw[0], root->tw[1], out, ctx->P.n, ctx->P.pbs_logB, root->p0inv_mod_p1);
    return hipGetLastError();
}

template <typename C, int MODE = 0>
hipError_t launch_pbs64_c(helm_si_ctx *ctx, const Pbs64Job *jobs, int64_t count, const uint64_t *small,
                          const uint64_t *luts, uint64_t *out, const double *key = nullptr, int n_steps = -1,
                          int logB = 0, size_t key_stride = 0, int key_first = 0, int *per_cu = nullptr)
{
    static std::atomic<bool> attr_done[64]; // (rank threads of one process launch concurrently)
    auto kern = k_pbs64<C, MODE>;
    if (!attr_done[ctx->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::BYTES);
        if (e != hipSuccess) return e;
        attr_done[ctx->device & 63] = true;
    }
    if (per_cu) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, 64 * C::NW, C::BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(64 * C::NW), C::BYTES, ctx->stream, jobs, small, luts,
                       key ? key : ctx->bsk, ctx->tw[0], ctx->tw[1], out, n_steps >= 0 ? n_steps : ctx->P.n,
                       logB ? logB : ctx->P.pbs_logB, ctx->p0inv_mod_p1, key_stride, key_first);
#ifdef HELM_WIDE_STAMPS
    {
        unsigned long long v[8 * 8];
        (void)hipStreamSynchronize(ctx->stream);
        if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_stamps64), sizeof(v)) == hipSuccess)
            for (int w = 0; w < C::NW; w++)
                fprintf(stderr, "[stamps k_pbs64: prep | fwd | products | bar1 | sum+bar2 | inverse | crt+bars] wave %d: %llu %llu %llu %llu %llu %llu %llu cycles/step\n",
                        w, v[w * 8] / ctx->P.n, v[w * 8 + 1] / ctx->P.n, v[w * 8 + 2] / ctx->P.n, v[w * 8 + 3] / ctx->P.n,
                        v[w * 8 + 4] / ctx->P.n, v[w * 8 + 5] / ctx->P.n, v[w * 8 + 6] / ctx->P.n);
    }
#endif
    return hipGetLastError();
}

template <typename C, bool MB>
hipError_t launch_pbs64s_c(helm_si_ctx *ctx, const Pbs64Job *jobs, int64_t count, const uint64_t *small,
                           const uint64_t *luts, uint64_t *out, int *per_cu = nullptr)
{
    static std::atomic<bool> attr_done[64]; // (rank threads of one process launch concurrently)
    auto kern = k_pbs64s<C, MB>;
    if (!attr_done[ctx->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::BYTES);
        if (e != hipSuccess) return e;
        attr_done[ctx->device & 63] = true;
    }
    if (per_cu) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, 64 * C::NW, C::BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(64 * C::NW), C::BYTES, ctx->stream, jobs, small, luts,
                       ctx->bsk_split, ctx->tw_sub, ctx->tw[0], ctx->tw[1], out, ctx->P.n, ctx->P.pbs_logB,
                       ctx->p0inv_mod_p1, ctx->group, ctx->expo, ctx->psi_pow);
#ifdef HELM_WIDE_STAMPS
    {
        unsigned long long v[8 * 8];
        (void)hipStreamSynchronize(ctx->stream);
        if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_stamps64), sizeof(v)) == hipSuccess)
            for (int w = 0; w < C::NW; w++) {
                fprintf(stderr, "[stamps k_pbs64s: digits | bar1 | fwd | products | sum+bars | inverse | halves+bars | crt+bars] wave %d:", w);
                for (int q = 0; q < 8; q++) fprintf(stderr, " %llu", v[w * 8 + q] / ctx->P.n);
                fprintf(stderr, " cycles/step\n");
            }
    }
#endif
    return hipGetLastError();
}

// Two translation units from this one source (Makefile).  The compiler's max-ILP scheduling strategy
// (-mllvm -amdgpu-sched-strategy=max-ilp; there is no per-function switch) is worth +3.7 % on k_pbs64k and +2.0 % on the
// multi-bit k_pbs64s, and costs the classical k_pbs64s 0.8 % and its two-level build 1.8 % (same box, alternating,
// identical ciphertexts: profiles/r03/si_kernel_experiments.txt (15)).  So the launchers of the first two are compiled a
// second time with -DHELM_SI_TU=1 under that strategy - nothing else of the file is - and the main unit
// (-DHELM_SI_SPLIT_TU=1) calls them through this one function.  Without the two macros the file is a single unit as before.
} // namespace
__attribute__((visibility("hidden"))) hipError_t helm_si_tu1_launch_pbs64(helm_si_ctx *ctx, const void *jobs, int64_t count,
                                                                        const uint64_t *small, const uint64_t *luts,
                                                                        uint64_t *out, int *per_cu);
#if HELM_SI_TU == 1
hipError_t helm_si_tu1_launch_pbs64(helm_si_ctx *ctx, const void *jobs_v, int64_t count, const uint64_t *small,
                                    const uint64_t *luts, uint64_t *out, int *per_cu)
{
    const Pbs64Job *jobs = static_cast<const Pbs64Job *>(jobs_v);
    const helm_si_params &P = ctx->P;
    if (P.k == 3) return launch_pbs64k_c<Pbs64kCfg<9, 3>>(ctx, jobs, count, small, luts, out, per_cu);
    if (P.k == 2) return launch_pbs64k_c<Pbs64kCfg<9, 2>>(ctx, jobs, count, small, luts, out, per_cu);
    if (ctx->use_split && ctx->group > 1) {
        if (ctx->logN == 10) return launch_pbs64s_c<Pbs64sCfg<10>, true>(ctx, jobs, count, small, luts, out, per_cu);
        if (ctx->logN == 11) return launch_pbs64s_c<Pbs64sCfg<11>, true>(ctx, jobs, count, small, luts, out, per_cu);
    }
    return hipErrorInvalidValue;
}
#endif
#if HELM_SI_TU == 0 // ==== everything below: the main unit only ===============================================
namespace {

hipError_t launch_pbs64(helm_si_ctx *ctx, const Pbs64Job *jobs, int64_t count, const uint64_t *small,
                        const uint64_t *luts, uint64_t *out, int *per_cu = nullptr)
{
    const helm_si_params &P = ctx->P;
#if HELM_SI_SPLIT_TU
    if (P.k == 3 || P.k == 2 || (ctx->use_split && ctx->group > 1))
        return helm_si_tu1_launch_pbs64(ctx, jobs, count, small, luts, out, per_cu);
#endif
    if (P.k == 3) return launch_pbs64k_c<Pbs64kCfg<9, 3>>(ctx, jobs, count, small, luts, out, per_cu);
    if (P.k == 2) return launch_pbs64k_c<Pbs64kCfg<9, 2>>(ctx, jobs, count, small, luts, out, per_cu);
    if (ctx->use_split) {
        if (ctx->group > 1) {
            if (ctx->logN == 10) return launch_pbs64s_c<Pbs64sCfg<10>, true>(ctx, jobs, count, small, luts, out, per_cu);
            if (ctx->logN == 11) return launch_pbs64s_c<Pbs64sCfg<11>, true>(ctx, jobs, count, small, luts, out, per_cu);
        }
        if (P.pbs_l == 1) {
            if (ctx->logN == 10) return launch_pbs64s_c<Pbs64sCfg<10>, false>(ctx, jobs, count, small, luts, out, per_cu);
            if (ctx->logN == 11) return launch_pbs64s_c<Pbs64sCfg<11>, false>(ctx, jobs, count, small, luts, out, per_cu);
        } else {
            if (ctx->logN == 10) return launch_pbs64s_c<Pbs64sCfg<10, 2>, false>(ctx, jobs, count, small, luts, out, per_cu);
            if (ctx->logN == 11) return launch_pbs64s_c<Pbs64sCfg<11, 2>, false>(ctx, jobs, count, small, luts, out, per_cu);
        }
    }
#define PBS64_CASE(LN, LV) \
    if (ctx->logN == LN && P.pbs_l == LV) return launch_pbs64_c<Pbs64Cfg<LN, LV>>(ctx, jobs, count, small, luts, out, nullptr, -1, 0, 0, 0, per_cu);
    PBS64_CASE(9, 1) PBS64_CASE(9, 2) PBS64_CASE(10, 1) PBS64_CASE(10, 2) PBS64_CASE(11, 1) PBS64_CASE(11, 2)
#undef PBS64_CASE
    return hipErrorInvalidValue;
}

// A keyswitching key on the device: in_dim input words (+ body) -> out_dim mask words + body.  The context's own
// key (big -> small) and the WoP-PBS path's extra keys go through the same kernels.
struct KsKey {
    const uint64_t *key = nullptr; // [in_dim][l][out_dim+1]
    const int8_t *planes = nullptr; // byte planes for the matrix-core kernel, or null
    int in_dim = 0, out_dim = 0, l = 0, logB = 0, kchunks = 0, ctiles = 0;
};

hipError_t launch_ks64_key(helm_si_ctx *ctx, const KsKey &K, const Ks64Job *jobs, int64_t count, const uint64_t *big,
                           uint64_t *out)
{
    // `out`: rows of out_dim+1 words (job g writes row jobs[g].out_row)
    struct { int n, ks_l, ks_logB; } P{K.out_dim, K.l, K.logB};
    const int kN = K.in_dim;
    // narrow batches stay on the vector-ALU kernel, whose key rows are split over workgroup slices: a matrix-core
    // wave walks the whole key column by column tile (0.23 ms whatever the width; vector ALU: 0.11 ms for 64, 0.37 ms
    // for 256 ciphertexts under PARAM_MESSAGE_2_CARRY_2)
    if (K.planes && ctx->ks_mfma && count >= 160) {
        const int64_t padded = (count + 63) / 64 * 64;
        const int LP = P.ks_l <= 1 ? 1 : P.ks_l <= 2 ? 2 : P.ks_l <= 4 ? 4 : 8;
        if (ctx->d_ksdig.ensure((size_t)padded * kN * LP) || ctx->d_ksdsum.ensure((size_t)padded) || ctx->d_ksbody.ensure((size_t)padded))
            return hipErrorOutOfMemory;
#define KSD_CASE(LV)                                                                                                  \
    case LV:                                                                                                          \
        hipLaunchKernelGGL(k_ks64_digits<LV>, dim3((unsigned)padded), dim3(256), 0, ctx->stream, jobs, big, ctx->d_ksdig.p, \
                           ctx->d_ksdsum.p, ctx->d_ksbody.p, kN, P.ks_logB, (int)count, K.kchunks);                 \
        break;
        switch (P.ks_l) {
            KSD_CASE(1) KSD_CASE(2) KSD_CASE(3) KSD_CASE(4) KSD_CASE(5) KSD_CASE(6) KSD_CASE(7) KSD_CASE(8)
        default: return hipErrorInvalidValue;
        }
#undef KSD_CASE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ks64_mfma, dim3((unsigned)(padded / 64), (unsigned)K.ctiles), dim3(64), 0, ctx->stream, jobs,
                           ctx->d_ksdig.p, ctx->d_ksdsum.p, ctx->d_ksbody.p, K.planes, out, P.n, (int)count,
                           K.kchunks, K.ctiles);
        return hipGetLastError();
    }
    const unsigned gx = (unsigned)((count + 3) / 4), gy = (unsigned)((P.n + 1 + 255) / 256);
    int slices = 1;
    while (slices < 16 && (int64_t)gx * gy * slices < 2 * (int64_t)ctx->n_cus && kN / (slices * 2) >= 64) slices *= 2;
    const int t_chunk = (kN + slices - 1) / slices;
    dim3 grid(gx, gy, (unsigned)slices);
    const size_t lds = (size_t)t_chunk * P.ks_l * sizeof(uint32_t);
    if (slices > 1) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)count * ((size_t)P.n + 1) * sizeof(uint64_t), ctx->stream);
        if (e != hipSuccess) return e;
    }
#define KS_CASE(LV)                                                                                                 \
    case LV: {                                                                                                      \
        static bool done[64] = {false}; /* per device: the attribute belongs to the device's code object */         \
        if (!done[ctx->device & 63]) {                                                                              \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_keyswitch64<LV>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);             \
            if (e != hipSuccess) return e;                                                                          \
            done[ctx->device & 63] = true;                                                                          \
        }                                                                                                           \
        hipLaunchKernelGGL(k_keyswitch64<LV>, grid, dim3(256), lds, ctx->stream, jobs, big, K.key, out, P.n, kN,   \
                           P.ks_logB, (int)count, t_chunk);                                                         \
        break;                                                                                                      \
    }
    switch (P.ks_l) {
        KS_CASE(1) KS_CASE(2) KS_CASE(3) KS_CASE(4) KS_CASE(5) KS_CASE(6) KS_CASE(7) KS_CASE(8)
    default: return hipErrorInvalidValue;
    }
#undef KS_CASE
    return hipGetLastError();
}

hipError_t launch_ks64(helm_si_ctx *ctx, const Ks64Job *jobs, int64_t count, const uint64_t *big, uint64_t *out)
{
    KsKey K;
    K.key = ctx->ksk;
    K.planes = ctx->ksk_planes;
    K.in_dim = ctx->P.k * ctx->P.N;
    K.out_dim = ctx->P.n;
    K.l = ctx->P.ks_l;
    K.logB = ctx->P.ks_logB;
    K.kchunks = ctx->ks_kchunks;
    K.ctiles = ctx->ks_ctiles;
    return launch_ks64_key(ctx, K, jobs, count, big, out);
}

int check_rows(const helm_si_wires *w, const int32_t *idx, int64_t count, bool allow_neg)
{
    for (int64_t i = 0; i < count; i++)
        if (idx[i] >= w->n_rows || (idx[i] < 0 && !(allow_neg && idx[i] == -1)))
            return fail(HELM_ERR_INVALID, "row index " + std::to_string(idx[i]) + " out of range at position " +
                                              std::to_string(i));
    return 0;
}

// Per-context scratch is refilled from pageable host memory by every call; such a copy is not
// guaranteed to queue behind kernels still reading the previous contents, so callers drain the
// stream (drain()) before their first upload.
int drain(helm_si_ctx *ctx)
{
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

template <typename T>
int upload(helm_si_ctx *ctx, DevBuf<T> &buf, const T *host, size_t n)
{
    if (buf.ensure(n)) return fail(HELM_ERR_OOM, "device scratch");
    HIP_TRY(hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

// ---- call slots (see helm_si_ctx) -------------------------------------------------------------------------
int slot_begin(helm_si_ctx *ctx, size_t need, helm_si_ctx::CallSlot **out)
{
    helm_si_ctx::CallSlot &S = ctx->slots[ctx->next_slot];
    ctx->next_slot = (ctx->next_slot + 1) % 3;
    if (!S.done) HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
    if (S.busy) HIP_TRY(hipEventSynchronize(S.done)); // two calls later at the earliest: normally long complete
    S.busy = false;
    need += 4096;
    if (need > S.cap) {
        if (S.host) (void)hipHostFree(S.host);
        if (S.dev) (void)hipFree(S.dev);
        S.host = S.dev = nullptr;
        S.cap = 0;
        const size_t want = std::max(need * 2, (size_t)1 << 20);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.host), want, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.dev), want));
        S.cap = want;
    }
    S.used = 0;
    *out = &S;
    return 0;
}
template <typename T> T *slot_put(helm_si_ctx::CallSlot &S, const T *src, size_t n)
{
    S.used = (S.used + 255) / 256 * 256;
    std::memcpy(S.host + S.used, src, n * sizeof(T));
    T *d = reinterpret_cast<T *>(S.dev + S.used);
    S.used += n * sizeof(T);
    return d;
}
int slot_flush(helm_si_ctx *ctx, helm_si_ctx::CallSlot &S)
{
    if (S.used) HIP_TRY(hipMemcpyAsync(S.dev, S.host, S.used, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}
int slot_end(helm_si_ctx *ctx, helm_si_ctx::CallSlot &S)
{
    HIP_TRY(hipEventRecord(S.done, ctx->stream));
    S.busy = true;
    return 0;
}
// the look-up tables of a call: resident already (same words as the last upload) or uploaded after a drain
int luts_resident(helm_si_ctx *ctx, const uint64_t *luts_host, size_t words)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ words;
    for (size_t i = 0; i < words; i++) h = (h ^ luts_host[i]) * 0x100000001B3ull + (h >> 29);
    // the hash only says "probably": the words themselves decide (a collision would bootstrap with the previous call's
    // tables and return wrong ciphertexts without an error; the WoP-PBS path swaps tables on one context all the time)
    if (ctx->d_luts.p && ctx->luts_words == words && ctx->luts_hash == h && ctx->luts_host.size() == words &&
        std::memcmp(ctx->luts_host.data(), luts_host, words * sizeof(uint64_t)) == 0)
        return 0;
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // a running bootstrap may still read the old tables
    if (ctx->d_luts.ensure(words)) return fail(HELM_ERR_OOM, "look-up tables");
    HIP_TRY(hipMemcpy(ctx->d_luts.p, luts_host, words * sizeof(uint64_t), hipMemcpyHostToDevice));
    ctx->luts_host.assign(luts_host, luts_host + words);
    ctx->luts_hash = h;
    ctx->luts_words = words;
    return 0;
}

// keyswitch + bootstrap of `count` rows of `src` (big) into rows of `dst` (big)
int apply_luts_device(helm_si_ctx *ctx, const uint64_t *src, uint64_t *dst, const std::vector<Ks64Job> &ks,
                      const std::vector<Pbs64Job> &pbs, const uint64_t *luts_host, int64_t n_luts,
                      const KsKey *other_key = nullptr) // other_key: keyswitch from another big key (WoP-PBS path)
{
    const helm_si_params &P = ctx->P;
    if (!ctx->have_bsk || !(ctx->have_ksk || other_key))
        return fail(HELM_ERR_STATE, "bootstrapping / keyswitching key not loaded");
    const int64_t count = (int64_t)pbs.size();
    if (count == 0) return 0;
    if (ctx->d_small.cap < (size_t)count * ((size_t)P.n + 1)) {
        if (int rc = drain(ctx)) return rc; // growing the scratch frees the old one
        if (ctx->d_small.ensure((size_t)count * ((size_t)P.n + 1))) return fail(HELM_ERR_OOM, "small-LWE scratch");
    }
    if (int rc = luts_resident(ctx, luts_host, (size_t)n_luts * P.N)) return rc;
    helm_si_ctx::CallSlot *S = nullptr;
    if (int rc = slot_begin(ctx, ks.size() * sizeof(Ks64Job) + pbs.size() * sizeof(Pbs64Job), &S)) return rc;
    const Ks64Job *d_ks = slot_put(*S, ks.data(), ks.size());
    const Pbs64Job *d_pbs = slot_put(*S, pbs.data(), pbs.size());
    if (int rc = slot_flush(ctx, *S)) return rc;
    {
        Timed t(ctx, &ctx->ev_ks);
        if (other_key) HIP_TRY(launch_ks64_key(ctx, *other_key, d_ks, count, src, ctx->d_small.p));
        else HIP_TRY(launch_ks64(ctx, d_ks, count, src, ctx->d_small.p));
    }
    ctx->tacc.ks_launches++;
    ctx->tacc.ks_count += count;
    {
        Timed t(ctx, &ctx->ev_pbs);
        HIP_TRY(launch_pbs64(ctx, d_pbs, count, ctx->d_small.p, ctx->d_luts.p, dst));
    }
    if (int rc = slot_end(ctx, *S)) return rc;
    ctx->tacc.pbs_launches++;
    ctx->tacc.pbs_count += count;
    return 0;
}

// Multi-bit: which power of psi is the evaluation point of each spectrum position, in the order the key
// words (and the transform outputs of k_pbs64s) are held?  Transform the polynomial X through the key
// conversion kernel (one "key" of one polynomial) and look the values up among the powers of psi.
int probe_spectrum_positions(helm_si_ctx *ctx)
{
    const int N = ctx->P.N, H = N / 2;
    std::vector<uint64_t> xpoly((size_t)N, 0);
    xpoly[1] = 1;
    // freed on every return path
    struct Tmp {
        void *p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } t_x, t_out;
    HIP_TRY(hipMalloc(&t_x.p, sizeof(uint64_t) * N));
    HIP_TRY(hipMalloc(&t_out.p, sizeof(double) * 4 * H));
    uint64_t *d_x = static_cast<uint64_t *>(t_x.p);
    double *d_out = static_cast<double *>(t_out.p);
    HIP_TRY(hipMemcpy(d_x, xpoly.data(), sizeof(uint64_t) * N, hipMemcpyHostToDevice));
    // one polynomial, K1 = 1: blocks (poly 0, half h) write d_out[(f * 2 + h) * N/2 ...] scaled by 1 (n_inv = 1)
#define PROBE(LN)                                                                                                        \
    if (ctx->logN == LN) {                                                                                              \
        hipLaunchKernelGGL((k_bsk_convert64s<F0, LN>), dim3(2), dim3(64), 0, ctx->stream, d_x, d_out, ctx->tw[0],        \
                           ctx->tw_sub, 1.0, ctx->two32[0], 1, 0, 1);                                                    \
        hipLaunchKernelGGL((k_bsk_convert64s<F1, LN>), dim3(2), dim3(64), 0, ctx->stream, d_x, d_out, ctx->tw[1],        \
                           ctx->tw_sub + (size_t)2 * H, 1.0, ctx->two32[1], 1, 1, 1);                                    \
    }
    PROBE(10) PROBE(11)
#undef PROBE
    HIP_TRY(hipGetLastError());
    std::vector<double> val((size_t)4 * H), pw((size_t)4 * N);
    HIP_TRY(hipMemcpyAsync(val.data(), d_out, sizeof(double) * 4 * H, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(pw.data(), ctx->psi_pow, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<uint16_t> expo((size_t)2 * H);
    for (int f = 0; f < 2; f++) {
        std::map<int64_t, int> where;
        for (int t = 0; t < 2 * N; t++) where[(int64_t)pw[(size_t)f * 2 * N + t]] = t;
        for (int q = 0; q < 2 * H; q++) {
            auto it = where.find((int64_t)val[(size_t)f * 2 * H + q]);
            if (it == where.end() || !(it->second & 1)) return fail(HELM_ERR_HIP, "multi-bit: spectrum position probe failed");
            if (f == 0) expo[(size_t)q] = (uint16_t)it->second;
            else if (expo[(size_t)q] != (uint16_t)it->second)
                return fail(HELM_ERR_HIP, "multi-bit: the two fields disagree on a spectrum position");
        }
    }
    { // the structure pbs64s_mb_body relies on: expo(h, lane, e) = expo(h, lane, 0) + (2N/EH) rev(e)  (mod 2N)
        const int EH = H / 64;
        int loge = 0;
        while ((1 << loge) < EH) loge++;
        for (int hh = 0; hh < 2; hh++)
            for (int e = 0; e < EH; e++)
                for (int lane = 0; lane < 64; lane++) {
                    const int v = expo[(size_t)hh * H + ((size_t)(e >> 1) * 64 + lane) * 2 + (e & 1)];
                    const int v0 = expo[(size_t)hh * H + (size_t)lane * 2];
                    if (v != ((v0 + (2 * N / EH) * bitrev_c(e, loge)) & (2 * N - 1)))
                        return fail(HELM_ERR_HIP, "multi-bit: unexpected order of the spectrum positions");
                }
    }
    if (const char *path = getenv("HELM_SI_DUMP_EXPO")) { // diagnostic: the probed table, [2 halves][N/2] u16
        if (FILE *fp = fopen(path, "wb")) {
            fwrite(expo.data(), sizeof(uint16_t), expo.size(), fp);
            fclose(fp);
        }
    }
    HIP_TRY(hipMalloc(&ctx->expo, expo.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(ctx->expo, expo.data(), expo.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return 0;
}

// all-gather of `rows` big-LWE rows per rank into x_gather, on the context's stream
int si_exchange(helm_si_ctx *ctx, int64_t rows)
{
    if (ctx->x_comm)
        return helm_comm_all_gather(ctx->x_comm, ctx->x_slot(rows), ctx->x_gather,
                                    (size_t)rows * ((size_t)ctx->P.k * ctx->P.N + 1) * sizeof(uint64_t), ctx->stream);
    if (int rc = ctx->x_fn(ctx->x_user, rows))
        return fail(HELM_ERR_STATE, "exchange callback failed with " + std::to_string(rc));
    return 0;
}

// helm_si_apply_luts with the batch sharded over the ranks of helm_si_set_exchange(): identical
// ciphertexts to the unsharded call (every bootstrap is independent and deterministic)
int apply_luts_sharded(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int32_t *lut_idx,
                       const int32_t *out_idx, int64_t count, const uint64_t *luts, int64_t n_luts)
{
    const int dim = ctx->P.k * ctx->P.N;
    const int64_t world = ctx->x_world;
    for (int64_t base = 0; base < count;) {
        const int64_t per = std::min(count - base, ctx->x_cap * world);
        const int64_t rows = (per + world - 1) / world;
        const int64_t lo = std::min(base + per, base + ctx->x_rank * rows), hi = std::min(base + per, lo + rows);
        std::vector<Ks64Job> ks((size_t)(hi - lo));
        std::vector<Pbs64Job> pbs((size_t)(hi - lo));
        for (int64_t g = lo; g < hi; g++) {
            ks[(size_t)(g - lo)] = Ks64Job{in_idx[g], (int32_t)(g - lo)};
            pbs[(size_t)(g - lo)] = Pbs64Job{(int32_t)(g - lo), lut_idx[g], (int32_t)(g - lo), 0};
        }
        if (int rc = apply_luts_device(ctx, w->d, ctx->x_slot(rows), ks, pbs, luts, n_luts)) return rc;
        // the collective (the library's ncclAllGather, or the caller's callback) on this context's stream: every
        // rank calls it, also one whose chunk is empty
        if (int rc = si_exchange(ctx, rows)) return rc;
        if (int rc = drain(ctx)) return rc;
        if (int rc = upload(ctx, ctx->d_idx2, out_idx + base, (size_t)per)) return rc;
        // gathered row q * rows + i is gate base + q * rows + i: the chunks are contiguous
        hipLaunchKernelGGL(k_rows64, dim3((unsigned)per), dim3(256), 0, ctx->stream, ctx->x_gather,
                           (const int32_t *)nullptr, w->d, ctx->d_idx2.p, dim);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->x_batches++;
        ctx->x_rows += rows * world;
        base += per;
    }
    return 0;
}

// The forward twiddle tables (bit-reversed powers of a primitive 2N-th root psi), 1/N, 2^32 and the CRT constant of one pair
// of fields.  pair 1 (the 46-bit fields): psi is chosen with psi^(N/4) = b, so that the first three table entries are b^2, b,
// b^3 - what the plain radix-4 top of a forward transform on digits multiplies by (fwd_top2_digits).
int setup_pair_tables(helm_si_ctx *ctx, int pair)
{
    const uint64_t pm[2] = {pair ? J0::P_U64 : F0::P_U64, pair ? J1::P_U64 : F1::P_U64};
    const uint64_t gen[2] = {pair ? J0::GEN : F0::GEN, pair ? J1::GEN : F1::GEN};
    const double b1[2] = {J0::B1, J1::B1}, b2[2] = {J0::B2, J1::B2}, b3[2] = {J0::B3, J1::B3};
    const int N = ctx->P.N, logN = ctx->logN;
    for (int f = 0; f < 2; f++) {
        uint64_t psi = powmod_u64(gen[f], (pm[f] - 1) / (2 * (uint64_t)N), pm[f]);
        if (pair) { // psi^(N/4) is one of the primitive eighth roots b, b^3, -b, -b^3: an odd power of psi puts it on b
            uint64_t pick = 0;
            for (uint64_t t = 1; t < 8 && !pick; t += 2)
                if (powmod_u64(powmod_u64(psi, t, pm[f]), (uint64_t)N / 4, pm[f]) == (uint64_t)b1[f]) pick = t;
            if (!pick) return fail(HELM_ERR_STATE, "internal: no 2N-th root of unity with psi^(N/4) = b");
            psi = powmod_u64(psi, pick, pm[f]);
        }
        std::vector<double> tf(N);
        uint64_t a = 1;
        for (int i = 0; i < N; i++) {
            tf[bitrev(i, logN)] = centred(a, pm[f]);
            a = mulmod_u64(a, psi, pm[f]);
        }
        if (pair && (tf[1] != b2[f] || tf[2] != b1[f] || tf[3] != b3[f]))
            return fail(HELM_ERR_STATE, "internal: the first twiddles are not the constants the kernels assume");
        ctx->n_inv[f] = centred(powmod_u64((uint64_t)N, pm[f] - 2, pm[f]), pm[f]);
        ctx->two32[f] = centred((1ull << 32) % pm[f], pm[f]);
        if (!ctx->tw[f]) HIP_TRY(hipMalloc(&ctx->tw[f], sizeof(double) * N));
        HIP_TRY(hipMemcpy(ctx->tw[f], tf.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    }
    ctx->p0inv_mod_p1 = centred(powmod_u64(pm[0] % pm[1], pm[1] - 2, pm[1]), pm[1]);
    ctx->pair = pair;
    return 0;
}


} // namespace

extern "C" {

int helm_si_ctx_create(int device_id, const helm_si_params *params, helm_si_ctx **out)
{
    if (!params || !out) return fail(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    const helm_si_params &P = *params;
    if (!si_supported(P))
        return fail(HELM_ERR_INVALID, "unsupported (k,N,pbs_l): built variants are k = 1, N in {512,1024,2048}, pbs_l in {1,2}; k in {2,3}, N = 512, pbs_l = 1");
    if (P.n < 1 || P.n > 1024) return fail(HELM_ERR_INVALID, "n must be in [1,1024]");
    // (pbs_logB <= 24: the kernels multiply digits by the field's fourth root of unity, 25 bits, without a reduction)
    if (P.pbs_logB < 2 || P.pbs_logB > 24 || P.pbs_logB * P.pbs_l > 31)
        return fail(HELM_ERR_INVALID, "bad PBS decomposition (pbs_logB <= 24 and pbs_logB * pbs_l <= 31: every tfhe shortint set)");
    if (P.ks_logB < 1 || P.ks_logB > 7 || P.ks_l < 1 || P.ks_l > 8 || P.ks_logB * P.ks_l > 63)
        return fail(HELM_ERR_INVALID, "bad keyswitch decomposition (ks_logB <= 7, ks_l <= 8)");
    const int t = P.message_modulus * P.carry_modulus;
    if (P.message_modulus < 2 || P.carry_modulus < 1 || (t & (t - 1)) || t > P.N / 2)
        return fail(HELM_ERR_INVALID, "message_modulus * carry_modulus must be a power of two <= N/2");
    const int group = P.grouping_factor > 1 ? P.grouping_factor : 1;
    if (P.grouping_factor < 0 || group > 3 || P.n % group)
        return fail(HELM_ERR_INVALID, "grouping_factor must be 0..3 and divide n");
    if (group > 1 && !(P.pbs_l == 1 && P.N >= 1024))
        return fail(HELM_ERR_INVALID, "multi-bit blind rotation is built for pbs_l = 1, N >= 1024 (every tfhe multi-bit set)");
    // exactness: |sum| <= (k+1) * l * N * (B/2) * 2^63 must stay below p0 * p1 / 2
    {
        const long double bound = (long double)(P.k + 1) * P.pbs_l * P.N * (long double)(1ull << (P.pbs_logB - 1)) *
                                  9223372036854775808.0L;
        if (bound * 1.001L >= (long double)F0::P * (long double)F1::P / 2)
            return fail(HELM_ERR_INVALID, "parameter set exceeds the two-prime NTT capacity");
        // per-field operands: digits must be far below p
        if ((double)(1ull << (P.pbs_logB - 1)) * 4 >= F1::P / 2) return fail(HELM_ERR_INVALID, "pbs_logB too large");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(HELM_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(HELM_ERR_NO_DEVICE, "device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(HELM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    helm_si_ctx *ctx = new (std::nothrow) helm_si_ctx();
    if (!ctx) return fail(HELM_ERR_OOM, "ctx");
    ctx->device = device_id;
    ctx->P = P;
    ctx->n_cus = prop.multiProcessorCount;
    while ((1 << ctx->logN) < P.N) ctx->logN++;
    ctx->delta = (1ull << 63) / (uint64_t)t;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    const uint64_t pm[2] = {F0::P_U64, F1::P_U64}, gen[2] = {F0::GEN, F1::GEN};
    const int N = P.N;
    if (int rc = setup_pair_tables(ctx, 0)) {
        (void)helm_si_ctx_destroy(ctx);
        return rc;
    }
    ctx->group = group;
    if (group > 1) { // the 2N powers of psi per field: monomial products in the transform domain
        std::vector<double> pw((size_t)4 * N);
        for (int f = 0; f < 2; f++) {
            const uint64_t psi = powmod_u64(gen[f], (pm[f] - 1) / (2 * (uint64_t)N), pm[f]);
            uint64_t a = 1;
            for (int i = 0; i < 2 * N; i++) {
                pw[(size_t)f * 2 * N + i] = centred(a, pm[f]);
                a = mulmod_u64(a, psi, pm[f]);
            }
        }
        HIP_TRY(hipMalloc(&ctx->psi_pow, pw.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(ctx->psi_pow, pw.data(), pw.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // eight-wave kernel (split transforms) where it exists: N >= 1024, one or two levels (HELM_SI_SPLIT=0: off)
    static_assert(Pbs64sCfg<11, 2>::BYTES <= 160 * 1024, "k_pbs64s<11, 2> must fit the LDS of a CU");
    ctx->use_split = ((P.pbs_l == 1 && P.pbs_logB <= 24) || (P.pbs_l == 2 && P.pbs_logB <= 15)) && N >= 1024 && P.k == 1;
    if (const char *v = getenv("HELM_HIP_KS_MFMA")) ctx->ks_mfma = atoi(v);
    if (const char *v = getenv("HELM_SI_SPLIT")) ctx->use_split = (ctx->use_split && atoi(v) != 0) || group > 1;
    if (ctx->use_split) {
        // half h of field f, stage with m' groups, group i': full table entry 2m' + h m' + i'
        std::vector<double> sub((size_t)4 * (N / 2), 0.0), full(N);
        for (int f = 0; f < 2; f++) {
            HIP_TRY(hipMemcpy(full.data(), ctx->tw[f], sizeof(double) * N, hipMemcpyDeviceToHost));
            for (int h = 0; h < 2; h++)
                for (int m = 1; m < N / 2; m <<= 1)
                    for (int i = 0; i < m; i++) sub[(size_t)(f * 2 + h) * (N / 2) + m + i] = full[2 * m + h * m + i];
        }
        HIP_TRY(hipMalloc(&ctx->tw_sub, sub.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(ctx->tw_sub, sub.data(), sub.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    *out = ctx;
    return 0;
}

int helm_si_ctx_fork(helm_si_ctx *primary, helm_si_ctx **out)
{
    if (!primary || !out) return fail(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (primary->lane_of) return fail(HELM_ERR_INVALID, "fork the primary context, not one of its lanes");
    if (!primary->have_bsk || !primary->have_ksk) return fail(HELM_ERR_STATE, "load the keys before forking a lane");
    HIP_TRY(hipSetDevice(primary->device));
    helm_si_ctx *ctx = new (std::nothrow) helm_si_ctx();
    if (!ctx) return fail(HELM_ERR_OOM, "ctx");
    ctx->device = primary->device;
    ctx->lane_of = primary;
    ctx->P = primary->P;
    ctx->logN = primary->logN;
    for (int f = 0; f < 2; f++) {
        ctx->tw[f] = primary->tw[f];
        ctx->n_inv[f] = primary->n_inv[f];
        ctx->two32[f] = primary->two32[f];
    }
    ctx->p0inv_mod_p1 = primary->p0inv_mod_p1;
    ctx->pair = primary->pair;
    ctx->bsk = primary->bsk;
    ctx->bsk_split = primary->bsk_split;
    ctx->tw_sub = primary->tw_sub;
    ctx->group = primary->group;
    ctx->expo = primary->expo;
    ctx->psi_pow = primary->psi_pow;
    ctx->use_split = primary->use_split;
    ctx->ksk = primary->ksk;
    ctx->ksk_planes = primary->ksk_planes;
    ctx->ks_kchunks = primary->ks_kchunks;
    ctx->ks_ctiles = primary->ks_ctiles;
    ctx->ks_mfma = primary->ks_mfma;
    ctx->have_bsk = ctx->have_ksk = true;
    ctx->delta = primary->delta;
    ctx->n_cus = primary->n_cus;
    ctx->audit_fn = primary->audit_fn; // a lane is audited like its primary
    ctx->audit_user = primary->audit_user;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(HELM_ERR_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return 0;
}

int helm_si_ctx_destroy(helm_si_ctx *ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto *l : {&ctx->ev_pbs, &ctx->ev_ks, &ctx->ev_lin})
        for (auto &p : *l) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto *w : ctx->child_wires) {
        (void)hipFree(w->d);
        w->d = nullptr;
        w->owner = nullptr;
    }
    if (!ctx->lane_of) { // a lane borrows its keys and tables
        (void)hipFree(ctx->tw[0]);
        (void)hipFree(ctx->tw[1]);
        (void)hipFree(ctx->bsk);
        (void)hipFree(ctx->bsk_split);
        (void)hipFree(ctx->tw_sub);
        (void)hipFree(ctx->expo);
        (void)hipFree(ctx->x_own);
        (void)hipFree(ctx->psi_pow);
        (void)hipFree(ctx->ksk);
        (void)hipFree(ctx->ksk_planes);
    }
    for (auto &S : ctx->slots) {
        if (S.host) (void)hipHostFree(S.host);
        if (S.dev) (void)hipFree(S.dev);
        if (S.done) (void)hipEventDestroy(S.done);
    }
    ctx->d_ksdig.release();
    ctx->d_ksdsum.release();
    ctx->d_ksbody.release();
    ctx->d_pbs.release();
    ctx->d_ks.release();
    ctx->d_small.release();
    ctx->d_luts.release();
    ctx->d_stage.release();
    ctx->d_body.release();
    ctx->d_idx.release();
    ctx->d_idx2.release();
    ctx->d_coef.release();
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return 0;
}

int helm_si_get_params(const helm_si_ctx *ctx, helm_si_params *out)
{
    if (!ctx || !out) return fail(HELM_ERR_INVALID, "null argument");
    *out = ctx->P;
    return 0;
}

int helm_si_field_bits(const helm_si_ctx *ctx)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null argument");
    return (ctx->lane_of ? ctx->lane_of : ctx)->pair ? 46 : 49;
}

int helm_si_set_stream(helm_si_ctx *ctx, void *hip_stream)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null ctx");
    if (int rc = helm_hip_runtime_guard_("helm_si_set_stream")) return rc; // the handle was made by the caller's HIP runtime
    // NULL is HIP's null (legacy default) stream - what torch.cuda.current_stream() is unless the caller
    // switched streams - NOT "back to the context's own stream": collectives the caller orders on that
    // stream must see the engine's kernels on it
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return 0;
}

int helm_si_set_priority(helm_si_ctx *ctx, int high)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null ctx");
    if (ctx->stream != ctx->own_stream) return 0; // the caller's stream: its priority is the caller's business
    if ((high != 0) == ctx->high_priority) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamSynchronize(ctx->own_stream));
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
    (void)hipStreamDestroy(ctx->own_stream);
    ctx->own_stream = ctx->stream = s;
    ctx->high_priority = high != 0;
    return 0;
}

int helm_si_sync(helm_si_ctx *ctx)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_load_bootstrap_key(helm_si_ctx *ctx, const uint64_t *bsk_std, size_t n_words)
{
    if (!ctx || !bsk_std) return fail(HELM_ERR_INVALID, "null argument");
    const helm_si_params &P = ctx->P;
    const size_t K1 = P.k + 1;
    const size_t n_ggsw = ctx->group > 1 ? (size_t)(P.n / ctx->group) << ctx->group : (size_t)P.n;
    const size_t polys = n_ggsw * P.pbs_l * K1 * K1;
    if (n_words != polys * P.N)
        return fail(HELM_ERR_INVALID, "bootstrapping key: expected " + std::to_string(polys * P.N) + " words, got " +
                                          std::to_string(n_words));
    HIP_TRY(hipSetDevice(ctx->device));
    struct Tmp {
        void *p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } t_std; // freed on every return path
    HIP_TRY(hipMalloc(&t_std.p, n_words * sizeof(uint64_t)));
    uint64_t *d_std = static_cast<uint64_t *>(t_std.p);
    if (!ctx->bsk && ctx->group == 1) HIP_TRY(hipMalloc(&ctx->bsk, n_words * 2 * sizeof(double)));
    if (P.k >= 2 && P.N == 512 && P.pbs_l == 1 && ctx->group == 1 && !ctx->lane_of) {
        // k_pbs64k contexts: the CRT pair follows the key at hand.  An exact product of a blind-rotation step is at most
        // B/2 x the largest l1-norm over the key polynomials that meet in one output column (or, transposed, in one row) - an
        // exact guarantee for this key and every input.  Below p p' / 2 of the 46-bit pair (and with digits of at most 17 bits,
        // whose products with b^3 stay exact doubles): FpJ / FpJ2, whose headroom drops stage 2's modular multiplications and
        // most recentrings (ntt_fp64.h); otherwise the 49-bit pair.  HELM_SI_FIELD=49 keeps the 49-bit pair.
        long double worst = 0;
        const size_t per_step = (size_t)P.pbs_l * K1 * K1; // src is [i][lev][r][c][N]
        std::vector<long double> l1(per_step);
        for (size_t i = 0; i < (size_t)P.n; i++) {
            for (size_t q = 0; q < per_step; q++) {
                const uint64_t *poly = bsk_std + (i * per_step + q) * P.N;
                long double sum = 0;
                for (int j = 0; j < P.N; j++) {
                    const int64_t v = (int64_t)poly[j];
                    sum += v < 0 ? -(long double)v : (long double)v;
                }
                l1[q] = sum;
            }
            for (size_t c = 0; c < K1; c++) {
                long double by_col = 0, by_row = 0;
                for (size_t q = 0; q < per_step; q++) {
                    if (q % K1 == c) by_col += l1[q];
                    if ((q / K1) % K1 == c) by_row += l1[q];
                }
                worst = std::max(worst, std::max(by_col, by_row));
            }
        }
        const long double key_bound = worst * (long double)(1ull << (P.pbs_logB - 1));
        // (margin 5 %: the CRT's quotient t = (r1 - r0) / p0 mod p1 comes out of mulmod within 0.512 p1, so the lifted value is
        // the exact integer as long as that stays below 0.488 p0 p1)
        int want = (key_bound * 1.05L < (long double)J0::P * (long double)J1::P / 2 && P.pbs_logB <= 18) ? 1 : 0;
        if (const char *v = getenv("HELM_SI_FIELD")) if (atoi(v) == 49) want = 0;
        if (want != ctx->pair) {
            HIP_TRY(hipStreamSynchronize(ctx->stream)); // nothing in flight may still read the other pair's tables
            ctx->have_bsk = false;
            if (int rc = setup_pair_tables(ctx, want)) return rc;
        }
    }
    HIP_TRY(hipMemcpyAsync(d_std, bsk_std, n_words * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->pair) { // (N = 512 only)
        hipLaunchKernelGGL((k_bsk_convert64<J0, 9>), dim3((unsigned)polys), dim3(64), 0, ctx->stream, d_std, ctx->bsk,
                           ctx->tw[0], ctx->n_inv[0], ctx->two32[0], (int)K1, P.pbs_l, 0);
        hipLaunchKernelGGL((k_bsk_convert64<J1, 9>), dim3((unsigned)polys), dim3(64), 0, ctx->stream, d_std, ctx->bsk,
                           ctx->tw[1], ctx->n_inv[1], ctx->two32[1], (int)K1, P.pbs_l, 1);
    }
#define CONV(LN)                                                                                                     \
    if (ctx->logN == LN && ctx->group == 1 && !ctx->pair) {                                                                                           \
        hipLaunchKernelGGL((k_bsk_convert64<F0, LN>), dim3((unsigned)polys), dim3(64), 0, ctx->stream, d_std, ctx->bsk, \
                           ctx->tw[0], ctx->n_inv[0], ctx->two32[0], (int)K1, P.pbs_l, 0);                           \
        hipLaunchKernelGGL((k_bsk_convert64<F1, LN>), dim3((unsigned)polys), dim3(64), 0, ctx->stream, d_std, ctx->bsk, \
                           ctx->tw[1], ctx->n_inv[1], ctx->two32[1], (int)K1, P.pbs_l, 1);                           \
    }
    CONV(9) CONV(10) CONV(11)
#undef CONV
    if (ctx->use_split) {
        if (!ctx->bsk_split) HIP_TRY(hipMalloc(&ctx->bsk_split, n_words * 2 * sizeof(double)));
#define CONVS(LN)                                                                                                       \
    if (ctx->logN == LN) {                                                                                              \
        hipLaunchKernelGGL((k_bsk_convert64s<F0, LN>), dim3((unsigned)(polys * 2)), dim3(64), 0, ctx->stream, d_std,     \
                           ctx->bsk_split, ctx->tw[0], ctx->tw_sub, ctx->n_inv[0], ctx->two32[0], (int)K1, 0, P.pbs_l);  \
        hipLaunchKernelGGL((k_bsk_convert64s<F1, LN>), dim3((unsigned)(polys * 2)), dim3(64), 0, ctx->stream, d_std,     \
                           ctx->bsk_split, ctx->tw[1], ctx->tw_sub + (size_t)2 * (P.N / 2), ctx->n_inv[1], ctx->two32[1], \
                           (int)K1, 1, P.pbs_l);                                                                         \
    }
        CONVS(10) CONVS(11)
#undef CONVS
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->group > 1 && !ctx->expo) {
        if (int rc = probe_spectrum_positions(ctx)) return rc;
    }
    ctx->have_bsk = true;
    return 0;
}

int helm_si_load_keyswitch_key(helm_si_ctx *ctx, const uint64_t *ksk, size_t n_words)
{
    if (!ctx || !ksk) return fail(HELM_ERR_INVALID, "null argument");
    const helm_si_params &P = ctx->P;
    const size_t want = (size_t)P.k * P.N * P.ks_l * ((size_t)P.n + 1);
    if (n_words != want)
        return fail(HELM_ERR_INVALID, "keyswitching key: expected " + std::to_string(want) + " words, got " +
                                          std::to_string(n_words));
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->ksk) HIP_TRY(hipMalloc(&ctx->ksk, want * sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(ctx->ksk, ksk, want * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    {
        // byte planes for the matrix-core keyswitch (levels padded to LP: the padding rows pair zero digits with -128)
        const int LP = P.ks_l <= 1 ? 1 : P.ks_l <= 2 ? 2 : P.ks_l <= 4 ? 4 : 8;
        const int R = P.k * P.N * LP;
        if (R % 64 == 0) {
            const int kchunks = R / 64, ctiles = (P.n + 1 + 15) / 16;
            const size_t frag_per_plane = (size_t)ctiles * kchunks * 64;
            std::vector<int8_t> planes(8 * frag_per_plane * 16, (int8_t)-128);
            const size_t krow = (size_t)P.n + 1;
            for (int ct = 0; ct < ctiles; ct++)
                for (int kc = 0; kc < kchunks; kc++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int c = ct * 16 + (lane & 15);
                        if (c > P.n) continue;
                        for (int j = 0; j < 16; j++) {
                            const int r = kc * 64 + 16 * (lane >> 4) + j, t = r / LP, lev = r % LP;
                            if (lev >= P.ks_l) continue;
                            const uint64_t w = ksk[((size_t)t * P.ks_l + lev) * krow + c];
                            for (int b = 0; b < 8; b++)
                                planes[((size_t)b * frag_per_plane + ((size_t)ct * kchunks + kc) * 64 + lane) * 16 + j] =
                                    (int8_t)((int)((w >> (8 * b)) & 255u) - 128);
                        }
                    }
            if (!ctx->ksk_planes) HIP_TRY(hipMalloc(&ctx->ksk_planes, planes.size()));
            HIP_TRY(hipMemcpy(ctx->ksk_planes, planes.data(), planes.size(), hipMemcpyHostToDevice));
            ctx->ks_kchunks = kchunks;
            ctx->ks_ctiles = ctiles;
        }
    }
    ctx->have_ksk = true;
    return 0;
}

int helm_si_wires_alloc(helm_si_ctx *ctx, int64_t n_rows, helm_si_wires **out)
{
    if (!ctx || !out || n_rows <= 0) return fail(HELM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    helm_si_wires *w = new (std::nothrow) helm_si_wires();
    if (!w) return fail(HELM_ERR_OOM, "wires");
    w->owner = ctx;
    w->n_rows = n_rows;
    const size_t bytes = (size_t)n_rows * ((size_t)ctx->P.k * ctx->P.N + 1) * sizeof(uint64_t);
    if (hipMalloc(&w->d, bytes) != hipSuccess) {
        delete w;
        return fail(HELM_ERR_OOM, "ciphertext table of " + std::to_string(bytes) + " bytes");
    }
    HIP_TRY(hipMemsetAsync(w->d, 0, bytes, ctx->stream));
    ctx->child_wires.push_back(w);
    *out = w;
    return 0;
}

int helm_si_wires_free(helm_si_ctx *ctx, helm_si_wires *w)
{
    if (!w) return 0;
    if (!w->owner) { // its context is gone and took the device memory with it
        delete w;
        return 0;
    }
    if (!ctx || w->owner != ctx) return fail(HELM_ERR_STATE, "table belongs to another context");
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(w->d);
    ctx->child_wires.erase(std::remove(ctx->child_wires.begin(), ctx->child_wires.end(), w), ctx->child_wires.end());
    delete w;
    return 0;
}

int helm_si_wires_upload(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, const uint64_t *lwe_host, int64_t count)
{
    if (!ctx || !w || !idx || !lwe_host || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(w, idx, count, false)) return rc;
    {
        std::vector<int32_t> sorted(idx, idx + count);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            return fail(HELM_ERR_INVALID, "upload names the same row twice");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain(ctx)) return rc;
    const int dim = ctx->P.k * ctx->P.N;
    if (int rc = upload(ctx, ctx->d_stage, lwe_host, (size_t)count * (dim + 1))) return rc;
    if (int rc = upload(ctx, ctx->d_idx, idx, (size_t)count)) return rc;
    hipLaunchKernelGGL(k_rows64, dim3((unsigned)count), dim3(256), 0, ctx->stream, ctx->d_stage.p, (const int32_t *)nullptr,
                       w->d, ctx->d_idx.p, dim);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_wires_download(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, uint64_t *lwe_host, int64_t count)
{
    if (!ctx || !w || !idx || !lwe_host || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(w, idx, count, false)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain(ctx)) return rc;
    const int dim = ctx->P.k * ctx->P.N;
    if (ctx->d_stage.ensure((size_t)count * (dim + 1))) return fail(HELM_ERR_OOM, "staging");
    if (int rc = upload(ctx, ctx->d_idx, idx, (size_t)count)) return rc;
    hipLaunchKernelGGL(k_rows64, dim3((unsigned)count), dim3(256), 0, ctx->stream, w->d, ctx->d_idx.p, ctx->d_stage.p,
                       (const int32_t *)nullptr, dim);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(lwe_host, ctx->d_stage.p, (size_t)count * (dim + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_wires_copy(helm_si_ctx *ctx, helm_si_wires *src, const int32_t *src_idx, helm_si_wires *dst,
                       const int32_t *dst_idx, int64_t count)
{
    if (!ctx || !src || !dst || !src_idx || !dst_idx || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, src) || !owns(ctx, dst)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(src, src_idx, count, false)) return rc;
    if (int rc = check_rows(dst, dst_idx, count, false)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain(ctx)) return rc;
    if (int rc = upload(ctx, ctx->d_idx, src_idx, (size_t)count)) return rc;
    if (int rc = upload(ctx, ctx->d_idx2, dst_idx, (size_t)count)) return rc;
    hipLaunchKernelGGL(k_rows64, dim3((unsigned)count), dim3(256), 0, ctx->stream, src->d, ctx->d_idx.p, dst->d,
                       ctx->d_idx2.p, ctx->P.k * ctx->P.N);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_wires_set_trivial(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, const uint64_t *value, int64_t count)
{
    if (!ctx || !w || !idx || !value || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(w, idx, count, false)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain(ctx)) return rc;
    std::vector<uint64_t> body((size_t)count);
    for (int64_t g = 0; g < count; g++) body[(size_t)g] = value[g] * ctx->delta;
    if (int rc = upload(ctx, ctx->d_body, body.data(), body.size())) return rc;
    if (int rc = upload(ctx, ctx->d_idx, idx, (size_t)count)) return rc;
    hipLaunchKernelGGL(k_set_trivial64, dim3((unsigned)count), dim3(256), 0, ctx->stream, ctx->d_idx.p, ctx->d_body.p, w->d,
                       ctx->P.k * ctx->P.N);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_bound_violations(helm_si_ctx *ctx, uint32_t counts[8], int reset)
{
    if (!ctx || !counts) return fail(HELM_ERR_INVALID, "null argument");
#ifdef HELM_CHECK_BOUNDS
    // this translation unit's own copy of the counters (ntt_fp64.h): every kernel of the 64-bit engine and of the WoP-PBS
    // path is compiled into it in the check build (one unit, no max-ILP split)
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_helm_bound_violations), 8 * sizeof(uint32_t)));
    if (reset) {
        const uint32_t zero[8] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_helm_bound_violations), zero, sizeof(zero)));
    }
    return 0;
#else
    (void)reset;
    return fail(HELM_ERR_STATE, "this library was not built with -DHELM_CHECK_BOUNDS (make libhelm_hip_check.so, HELM_HIP_LIB)");
#endif
}

int helm_si_set_audit(helm_si_ctx *ctx, helm_si_audit_fn fn, void *user)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null context");
    ctx->audit_fn = fn;
    ctx->audit_user = fn ? user : nullptr;
    return 0;
}

// rows idx[0 .. count) of the table on the host (idx < 0: a zero row), for the audit
static int audit_fetch(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, int64_t count, std::vector<uint64_t> &rows)
{
    const size_t brow = (size_t)ctx->P.k * ctx->P.N + 1;
    std::vector<int32_t> safe((size_t)count);
    for (int64_t g = 0; g < count; g++) safe[(size_t)g] = idx[g] < 0 ? 0 : idx[g];
    rows.assign((size_t)count * brow, 0);
    if (int rc = helm_si_wires_download(ctx, w, safe.data(), rows.data(), count)) return rc;
    for (int64_t g = 0; g < count; g++)
        if (idx[g] < 0) std::fill(rows.begin() + (size_t)g * brow, rows.begin() + (size_t)(g + 1) * brow, 0);
    return 0;
}

static int lincomb_impl(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int64_t *coef,
                        const int64_t *const_add, const int32_t *out_idx, int32_t terms, int64_t count);

int helm_si_lincomb(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int64_t *coef,
                    const int64_t *const_add, const int32_t *out_idx, int32_t terms, int64_t count)
{
    if (!ctx || !ctx->audit_fn || count <= 0) return lincomb_impl(ctx, w, in_idx, coef, const_add, out_idx, terms, count);
    if (!w || !in_idx || !coef || !out_idx || terms < 1) return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (int rc = check_rows(w, in_idx, count * terms, true)) return rc;
    std::vector<uint64_t> in_rows, out_rows;
    if (int rc = audit_fetch(ctx, w, in_idx, count * terms, in_rows)) return rc;
    if (int rc = lincomb_impl(ctx, w, in_idx, coef, const_add, out_idx, terms, count)) return rc;
    if (int rc = audit_fetch(ctx, w, out_idx, count, out_rows)) return rc;
    helm_si_audit_record rec{};
    rec.kind = 1;
    rec.count = count;
    rec.terms = terms;
    rec.in_rows = in_rows.data();
    rec.out_rows = out_rows.data();
    rec.in_idx = in_idx;
    rec.coef = coef;
    rec.const_add = const_add;
    if (int rc = ctx->audit_fn(ctx->audit_user, &rec))
        return fail(HELM_ERR_STATE, "helm_si_lincomb: the audit callback rejected the batch (" + std::to_string(rc) + ")");
    return 0;
}

static int lincomb_impl(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int64_t *coef,
                        const int64_t *const_add, const int32_t *out_idx, int32_t terms, int64_t count)
{
    if (!ctx || !w || !in_idx || !coef || !out_idx || terms < 1 || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(w, in_idx, count * terms, true)) return rc;
    if (int rc = check_rows(w, out_idx, count, false)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const int dim = ctx->P.k * ctx->P.N;
    std::vector<uint64_t> body((size_t)count, 0);
    if (const_add)
        for (int64_t g = 0; g < count; g++) body[(size_t)g] = (uint64_t)const_add[g] * ctx->delta;
    // sums are staged (rows 0..count-1) and then scattered, so that a gate may overwrite a
    // row another gate of the same call still reads
    if (ctx->d_stage.cap < (size_t)count * (dim + 1)) {
        if (int rc = drain(ctx)) return rc; // growing the staging area frees the old one
        if (ctx->d_stage.ensure((size_t)count * (dim + 1))) return fail(HELM_ERR_OOM, "staging");
    }
    helm_si_ctx::CallSlot *S = nullptr;
    if (int rc = slot_begin(ctx, (size_t)count * (8 + 4) + (size_t)count * terms * (4 + 8), &S)) return rc;
    const uint64_t *d_body = slot_put(*S, body.data(), body.size());
    const int32_t *d_in = slot_put(*S, in_idx, (size_t)count * terms);
    const int32_t *d_out = slot_put(*S, out_idx, (size_t)count);
    const int64_t *d_coef = slot_put(*S, coef, (size_t)count * terms);
    if (int rc = slot_flush(ctx, *S)) return rc;
    {
        Timed t(ctx, &ctx->ev_lin);
        hipLaunchKernelGGL(k_lincomb64, dim3((unsigned)count), dim3(256), 0, ctx->stream, d_in, d_coef, d_body,
                           (const int32_t *)nullptr, w->d, ctx->d_stage.p, terms, dim);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_rows64, dim3((unsigned)count), dim3(256), 0, ctx->stream, ctx->d_stage.p,
                           (const int32_t *)nullptr, w->d, d_out, dim);
        HIP_TRY(hipGetLastError());
    }
    return slot_end(ctx, *S);
}

int helm_si_make_lut(const helm_si_ctx *ctx, const uint64_t *f_values, uint64_t *tv)
{
    if (!ctx || !f_values || !tv) return fail(HELM_ERR_INVALID, "null argument");
    const int N = ctx->P.N, t = ctx->P.message_modulus * ctx->P.carry_modulus;
    const int box = N / t, half = box / 2;
    std::vector<uint64_t> acc((size_t)N);
    for (int v = 0; v < t; v++)
        for (int j = 0; j < box; j++) acc[(size_t)v * box + j] = f_values[v] * ctx->delta;
    for (int j = 0; j < half; j++) acc[(size_t)j] = 0ull - acc[(size_t)j];
    for (int j = 0; j < N; j++) tv[j] = acc[(size_t)((j + half) % N)]; // rotate_left(half)
    return 0;
}

int helm_si_apply_luts(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int32_t *lut_idx,
                       const int32_t *out_idx, int64_t count, const uint64_t *luts, int64_t n_luts)
{
    if (!ctx || !w || !in_idx || !lut_idx || !out_idx || !luts || count < 0 || n_luts <= 0)
        return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(w, in_idx, count, false)) return rc;
    if (int rc = check_rows(w, out_idx, count, false)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<Ks64Job> ks((size_t)count);
    std::vector<Pbs64Job> pbs((size_t)count);
    for (int64_t g = 0; g < count; g++) {
        if (lut_idx[g] < 0 || lut_idx[g] >= n_luts) return fail(HELM_ERR_INVALID, "lut_idx out of range");
        ks[(size_t)g] = Ks64Job{in_idx[g], (int32_t)g};
        pbs[(size_t)g] = Pbs64Job{(int32_t)g, lut_idx[g], out_idx[g], 0};
    }
    std::vector<uint64_t> audit_in, audit_out;
    if (ctx->audit_fn)
        if (int rc = audit_fetch(ctx, w, in_idx, count, audit_in)) return rc;
    int rc = 0;
    if (ctx->x_on && count >= ctx->x_min)
        rc = apply_luts_sharded(ctx, w, in_idx, lut_idx, out_idx, count, luts, n_luts);
    else // every keyswitch finishes (kernel boundary) before any bootstrap writes its output row
        rc = apply_luts_device(ctx, w->d, w->d, ks, pbs, luts, n_luts);
    if (rc || !ctx->audit_fn) return rc;
    if ((rc = audit_fetch(ctx, w, out_idx, count, audit_out))) return rc;
    helm_si_audit_record rec{};
    rec.kind = 0;
    rec.count = count;
    rec.n_luts = n_luts;
    rec.in_rows = audit_in.data();
    rec.out_rows = audit_out.data();
    rec.lut_idx = lut_idx;
    rec.luts = luts;
    if ((rc = ctx->audit_fn(ctx->audit_user, &rec)))
        return fail(HELM_ERR_STATE, "helm_si_apply_luts: the audit callback rejected the batch (" + std::to_string(rc) + ")");
    return 0;
}

int helm_si_set_exchange(helm_si_ctx *ctx, int32_t rank, int32_t world, int64_t min_batch, void *stage_dev,
                         void *gather_dev, int64_t capacity_rows, helm_si_exchange_fn fn, void *user)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null context");
    if (fn)
        if (int rc = helm_hip_runtime_guard_("helm_si_set_exchange")) return rc; // the caller's buffers and collective
    if (ctx->x_own) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->x_own);
        ctx->x_own = nullptr;
    }
    ctx->x_comm = nullptr;
    if (world < 1 || (world == 1 && !fn)) { // back to single-GPU evaluation
        ctx->x_world = 1;
        ctx->x_rank = 0;
        ctx->x_fn = nullptr;
        ctx->x_on = false;
        ctx->x_stage = ctx->x_gather = nullptr;
        return 0;
    }
    if (rank < 0 || rank >= world || min_batch < 1 || capacity_rows < 1 || !stage_dev || !gather_dev || !fn)
        return fail(HELM_ERR_INVALID, "helm_si_set_exchange: bad argument");
    ctx->x_on = true;
    ctx->x_rank = rank;
    ctx->x_world = world;
    ctx->x_min = min_batch;
    ctx->x_cap = capacity_rows;
    ctx->x_stage = static_cast<uint64_t *>(stage_dev);
    ctx->x_gather = static_cast<uint64_t *>(gather_dev);
    ctx->x_fn = fn;
    ctx->x_user = user;
    ctx->x_batches = ctx->x_rows = 0;
    return 0;
}

int helm_si_set_exchange_comm(helm_si_ctx *ctx, helm_comm *comm, int64_t min_batch, int64_t capacity_rows)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null context");
    if (!comm) return helm_si_set_exchange(ctx, 0, 1, 1, nullptr, nullptr, 1, nullptr, nullptr);
    if (min_batch < 1 || capacity_rows < 1) return fail(HELM_ERR_INVALID, "helm_si_set_exchange_comm: bad argument");
    int rank = 0, world = 0, dev = -1;
    if (int rc = helm_comm_info(comm, &rank, &world, &dev, nullptr)) return rc;
    if (dev != ctx->device) return fail(HELM_ERR_STATE, "helm_si_set_exchange_comm: the communicator lives on another device than the context");
    if (int rc = helm_si_set_exchange(ctx, 0, 1, 1, nullptr, nullptr, 1, nullptr, nullptr)) return rc; // releases an earlier buffer
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t brow = (size_t)ctx->P.k * ctx->P.N + 1;
    HIP_TRY(hipMalloc(&ctx->x_own, (size_t)capacity_rows * (size_t)world * brow * sizeof(uint64_t)));
    ctx->x_comm = comm;
    ctx->x_gather = ctx->x_own;
    ctx->x_stage = nullptr; // chunks go straight into the gather buffer (x_slot)
    ctx->x_rank = rank;
    ctx->x_world = world;
    ctx->x_min = min_batch;
    ctx->x_cap = capacity_rows;
    ctx->x_fn = nullptr;
    ctx->x_user = nullptr;
    ctx->x_on = true;
    ctx->x_batches = ctx->x_rows = 0;
    return 0;
}

int helm_si_exchange_world(const helm_si_ctx *ctx) { return ctx ? ctx->x_world : 1; }

int64_t helm_si_round_capacity(helm_si_ctx *ctx)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null context");
    if (ctx->round_capacity <= 0) {
        HIP_TRY(hipSetDevice(ctx->device));
        int per_cu = 0;
        HIP_TRY(launch_pbs64(ctx, nullptr, 0, nullptr, nullptr, nullptr, &per_cu));
        ctx->round_capacity = (int64_t)std::max(per_cu, 1) * ctx->n_cus;
    }
    return ctx->round_capacity;
}

int helm_si_exchange_stats(const helm_si_ctx *ctx, int64_t *batches, int64_t *rows)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null context");
    if (batches) *batches = ctx->x_batches;
    if (rows) *rows = ctx->x_rows;
    return 0;
}

int helm_si_eval_lut_level(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *arity, const int32_t *in_idx,
                           int32_t max_in, const uint64_t *table, const int32_t *out_idx, int64_t count)
{
    if (!ctx || !w || !arity || !in_idx || !table || !out_idx || max_in < 1 || count < 0)
        return fail(HELM_ERR_INVALID, "bad argument");
    if (!owns(ctx, w)) return fail(HELM_ERR_STATE, "table belongs to another context");
    if (count == 0) return 0;
    if (int rc = check_rows(w, out_idx, count, false)) return rc;
    // The gates of a level run concurrently (circuit.rs:1057: par_iter over the level; the reference's guarantee comes from
    // compute_levels, circuit.rs:174-239): a level in which a gate reads a row ANOTHER gate of the level writes, or in which
    // two gates write one row, is refused instead of evaluated in an unspecified order.  A gate may rewrite its own input row.
    {
        std::vector<std::pair<int32_t, int64_t>> writer((size_t)count);
        for (int64_t g = 0; g < count; g++) writer[(size_t)g] = {out_idx[g], g};
        std::sort(writer.begin(), writer.end());
        for (int64_t g = 1; g < count; g++)
            if (writer[(size_t)g].first == writer[(size_t)g - 1].first)
                return fail(HELM_ERR_INVALID, "gates " + std::to_string(writer[(size_t)g - 1].second) + " and " +
                                                  std::to_string(writer[(size_t)g].second) + " both write row " +
                                                  std::to_string(writer[(size_t)g].first) + " (write-after-write inside a level)");
        for (int64_t g = 0; g < count; g++) {
            const int ar = arity[g] < 0 ? 0 : (arity[g] > max_in ? max_in : arity[g]);
            for (int q = 0; q < std::max(ar, 1); q++) {
                const int32_t r = in_idx[(size_t)g * max_in + q];
                auto it = std::lower_bound(writer.begin(), writer.end(), std::make_pair(r, (int64_t)-1));
                if (it != writer.end() && it->first == r && it->second != g)
                    return fail(HELM_ERR_INVALID, "gate " + std::to_string(g) + " reads row " + std::to_string(r) + ", which gate " +
                                                      std::to_string(it->second) + " of the same level writes (read-after-write inside a level)");
            }
        }
    }
    const helm_si_params &P = ctx->P;
    const int t = P.message_modulus * P.carry_modulus;
    // ---- linear part of every gate: packed operand (LUT gates), copy or negation ---------
    std::vector<int32_t> lin_in((size_t)count * max_in, -1);
    std::vector<int64_t> lin_coef((size_t)count * max_in, 0);
    std::vector<int32_t> pbs_gate; // gates that bootstrap
    std::map<std::pair<int, uint64_t>, int32_t> lut_of;
    std::vector<uint64_t> luts;
    std::vector<int32_t> lut_idx;
    for (int64_t g = 0; g < count; g++) {
        const int ar = arity[g];
        if (ar < 0 || ar > max_in) return fail(HELM_ERR_INVALID, "gate " + std::to_string(g) + ": bad arity");
        const int32_t *in = in_idx + (size_t)g * max_in;
        for (int q = 0; q < std::max(ar, 1); q++)
            if (in[q] < 0 || in[q] >= w->n_rows)
                return fail(HELM_ERR_INVALID, "gate " + std::to_string(g) + ": operand row out of range");
        if (ar == 0) { // DFF / copy (circuit.rs:1063-1069)
            lin_in[(size_t)g * max_in] = in[0];
            lin_coef[(size_t)g * max_in] = 1;
        } else if (ar == 1) { // gates.rs:765-770
            lin_in[(size_t)g * max_in] = in[0];
            lin_coef[(size_t)g * max_in] = table[g] == 0 ? 1 : -1;
        } else {
            if ((1 << ar) > t)
                return fail(HELM_ERR_INVALID, "gate " + std::to_string(g) + ": " + std::to_string(ar) +
                                                  " inputs do not fit the plaintext space of " + std::to_string(t));
            for (int q = 0; q < ar; q++) { // gates.rs:773-778 (ar = 2: x * 2 + y)
                lin_in[(size_t)g * max_in + q] = in[q];
                lin_coef[(size_t)g * max_in + q] = (int64_t)1 << (ar - 1 - q);
            }
            const std::pair<int, uint64_t> key(ar, table[g]);
            auto it = lut_of.find(key);
            if (it == lut_of.end()) {
                std::vector<uint64_t> f((size_t)t, 0);
                for (int v = 0; v < t; v++) {
                    // arity 2: table[(x&1)*2 + (y&1)] with v = 2x + y (gates.rs:750-752); else table[v] & 1
                    const int idx = ar == 2 ? (((v >> 1) & 1) * 2 + (v & 1)) : (v & ((1 << ar) - 1));
                    f[(size_t)v] = (table[g] >> idx) & 1ull;
                }
                const int32_t id = (int32_t)(luts.size() / P.N);
                luts.resize(luts.size() + P.N);
                helm_si_make_lut(ctx, f.data(), luts.data() + (size_t)id * P.N);
                it = lut_of.emplace(key, id).first;
            }
            pbs_gate.push_back((int32_t)g);
            lut_idx.push_back(it->second);
        }
    }
    // bootstrapped gates first (pack, then KS + PBS in place), copies / negations after them:
    // the order the boolean engine uses, which only matters for a DFF that latches a wire
    // produced in its own (last) level
    std::vector<int32_t> sub_in, sub_out;
    std::vector<int64_t> sub_coef;
    auto gather = [&](bool want_pbs) {
        sub_in.clear();
        sub_out.clear();
        sub_coef.clear();
        for (int64_t g = 0; g < count; g++) {
            if ((arity[g] >= 2) != want_pbs) continue;
            sub_in.insert(sub_in.end(), lin_in.begin() + g * max_in, lin_in.begin() + (g + 1) * max_in);
            sub_coef.insert(sub_coef.end(), lin_coef.begin() + g * max_in, lin_coef.begin() + (g + 1) * max_in);
            sub_out.push_back(out_idx[g]);
        }
    };
    if (!pbs_gate.empty()) {
        gather(true);
        if (int rc = helm_si_lincomb(ctx, w, sub_in.data(), sub_coef.data(), nullptr, sub_out.data(), max_in,
                                     (int64_t)sub_out.size()))
            return rc;
        if (int rc = helm_si_apply_luts(ctx, w, sub_out.data(), lut_idx.data(), sub_out.data(), (int64_t)sub_out.size(),
                                        luts.data(), (int64_t)(luts.size() / P.N)))
            return rc;
    }
    gather(false);
    if (!sub_out.empty())
        if (int rc = helm_si_lincomb(ctx, w, sub_in.data(), sub_coef.data(), nullptr, sub_out.data(), max_in,
                                     (int64_t)sub_out.size()))
            return rc;
    return 0;
}

int helm_si_keyswitch_batch(helm_si_ctx *ctx, const uint64_t *in_big, uint64_t *out_small, int64_t count)
{
    if (!ctx || !in_big || !out_small || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    if (!ctx->have_ksk) return fail(HELM_ERR_STATE, "keyswitching key not loaded");
    if (count == 0) return 0;
    const helm_si_params &P = ctx->P;
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain(ctx)) return rc;
    const size_t row = (size_t)P.n + 1, brow = (size_t)P.k * P.N + 1;
    std::vector<Ks64Job> jobs((size_t)count);
    for (int64_t g = 0; g < count; g++) jobs[(size_t)g] = Ks64Job{(int32_t)g, (int32_t)g};
    if (int rc = upload(ctx, ctx->d_stage, in_big, (size_t)count * brow)) return rc;
    if (int rc = upload(ctx, ctx->d_ks, jobs.data(), jobs.size())) return rc;
    if (ctx->d_small.ensure((size_t)count * row)) return fail(HELM_ERR_OOM, "small-LWE scratch");
    {
        Timed t(ctx, &ctx->ev_ks);
        HIP_TRY(launch_ks64(ctx, ctx->d_ks.p, count, ctx->d_stage.p, ctx->d_small.p));
    }
    HIP_TRY(hipMemcpyAsync(out_small, ctx->d_small.p, (size_t)count * row * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_pbs_batch(helm_si_ctx *ctx, const uint64_t *in_small, const uint64_t *luts, int64_t n_luts,
                      const int32_t *lut_idx, uint64_t *out_big, int64_t count)
{
    if (!ctx || !in_small || !luts || !lut_idx || !out_big || count < 0 || n_luts <= 0)
        return fail(HELM_ERR_INVALID, "bad argument");
    if (!ctx->have_bsk) return fail(HELM_ERR_STATE, "bootstrapping key not loaded");
    if (count == 0) return 0;
    const helm_si_params &P = ctx->P;
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain(ctx)) return rc;
    const size_t row = (size_t)P.n + 1, brow = (size_t)P.k * P.N + 1;
    std::vector<Pbs64Job> jobs((size_t)count);
    for (int64_t g = 0; g < count; g++) {
        if (lut_idx[g] < 0 || lut_idx[g] >= n_luts) return fail(HELM_ERR_INVALID, "lut_idx out of range");
        jobs[(size_t)g] = Pbs64Job{(int32_t)g, lut_idx[g], (int32_t)g, 0};
    }
    if (int rc = upload(ctx, ctx->d_small, in_small, (size_t)count * row)) return rc;
    ctx->luts_words = 0; // the resident tables of the call slots are overwritten
    if (int rc = upload(ctx, ctx->d_luts, luts, (size_t)n_luts * P.N)) return rc;
    if (int rc = upload(ctx, ctx->d_pbs, jobs.data(), jobs.size())) return rc;
    if (ctx->d_stage.ensure((size_t)count * brow)) return fail(HELM_ERR_OOM, "staging");
    {
        Timed t(ctx, &ctx->ev_pbs);
        HIP_TRY(launch_pbs64(ctx, ctx->d_pbs.p, count, ctx->d_small.p, ctx->d_luts.p, ctx->d_stage.p));
    }
    ctx->tacc.pbs_launches++;
    ctx->tacc.pbs_count += count;
    HIP_TRY(hipMemcpyAsync(out_big, ctx->d_stage.p, (size_t)count * brow * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int helm_si_timing_enable(helm_si_ctx *ctx, int enable)
{
    if (!ctx) return fail(HELM_ERR_INVALID, "null ctx");
    ctx->timing = enable != 0;
    return 0;
}

int helm_si_get_timing(helm_si_ctx *ctx, helm_si_timing *out, int reset)
{
    if (!ctx || !out) return fail(HELM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>> &l, double &acc) {
        for (auto &p : l) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) acc += ms;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        l.clear();
    };
    drain(ctx->ev_pbs, ctx->tacc.pbs_ms);
    drain(ctx->ev_ks, ctx->tacc.ks_ms);
    drain(ctx->ev_lin, ctx->tacc.linear_ms);
    *out = ctx->tacc;
    if (reset) ctx->tacc = helm_si_timing{};
    return 0;
}

} // extern "C"

// the WoP-PBS wide-LUT path (include/helm_wopbs.h): same translation unit, it is built from the kernels and launch
// helpers above
#include "helm_wopbs.inc"
#endif // HELM_SI_TU == 0
