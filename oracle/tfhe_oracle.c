/*
 * tfhe_oracle.c — CPU restatement of the HELM gates-mode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under helm_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * What it restates.  HELM issues one call per gate into the `tfhe` crate
 * (reference src/gates.rs:254-275: ServerKey::{and,nand,or,nor,xor,xnor,mux,not});
 * the arithmetic lives in that third-party dependency, `tfhe = 0.4.1`
 * (reference Cargo.toml:18), whose source is NOT under /root/reference and
 * cannot be built here (no rustc/cargo, no network).  This file therefore
 * restates the published TFHE gate-bootstrap algorithm that crate implements
 * (linear step -> modulus switch -> blind rotate with CMUX/external product
 * over the GGSW bootstrapping key -> sample extract -> LWE keyswitch), using
 * the constants the reference itself pins: plaintext encoding +-1/8
 * (reference src/circuit.rs:29,33) and the decrypt rule "phase < 2^31 => true"
 * (src/circuit.rs:948), gate operand order (src/gates.rs:255-271), and the
 * parameter set of src/bin/helm.rs:141-146.
 *
 * PARITY STATUS: "parity unpinned" at ciphertext level — the reference's tests
 * never compare ciphertext bits (tests/gates_test.rs:82-107,
 * tests/circuit_test.rs:91-93 compare decrypted values only) and tfhe-rs uses
 * an f64 FFT whose rounding an exact method cannot reproduce.  The oracle is
 * pinned at the level the reference pins anything: decrypted truth tables and
 * decrypted netlist wires (tests/test_oracle_*.py).
 *
 * Exactness.  All polynomial products are computed EXACTLY over the integers
 * and reduced mod 2^32.  Two independent routes are provided and cross-checked
 * in the tests: (1) schoolbook negacyclic convolution in wrapping u32
 * arithmetic (obviously correct, O(N^2)); (2) a negacyclic NTT over the
 * Goldilocks prime 2^64-2^32+1 with centred lifting (scalar); (3) an fp64-FMA
 * NTT over a 49/51-bit prime, one gate per SIMD lane (AVX2 / AVX-512, picked at
 * run time: fp_route.inc) - the timed CPU baseline of bench.py.  All three are
 * exact, so they and the HIP path must agree bit for bit; route 3 runs boolean
 * parameter sets in a different prime field from the GPU's.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct {
    int32_t n;        /* small LWE dimension                       */
    int32_t k;        /* GLWE dimension                            */
    int32_t N;        /* polynomial size (power of two)            */
    int32_t pbs_l;    /* bootstrap decomposition levels            */
    int32_t pbs_logB; /* bootstrap decomposition base log          */
    int32_t ks_l;     /* keyswitch decomposition levels            */
    int32_t ks_logB;  /* keyswitch decomposition base log          */
} orc_params;

/* Gate op-codes: the order of `enum GateType`, reference src/gates.rs:23-45. */
enum {
    ORC_AND = 0, ORC_DFF = 1, ORC_LUT = 2, ORC_MUX = 3, ORC_NAND = 4, ORC_NOR = 5,
    ORC_NOT = 6, ORC_OR = 7, ORC_XNOR = 8, ORC_XOR = 9, ORC_BUF = 10,
    ORC_CONST_ONE = 11, ORC_CONST_ZERO = 12
};

#define PT_TRUE  ((u32)1 << 29)  /* +1/8, reference src/circuit.rs:29 */
#define PT_FALSE ((u32)7 << 29)  /* -1/8, reference src/circuit.rs:33 */

/* ------------------------------------------------------------------------- */
/* Linear pre-bootstrap step of each boolean gate.                            */
/* tfhe 0.4.1 boolean engine [dependency, restated]: AND l+r-1/8; OR l+r+1/8;  */
/* NAND -(l+r)+1/8; NOR -(l+r)-1/8; XOR 2(l+r+1/8); XNOR -2(l+r+1/8).           */
/* MUX(c,t,e) needs two bootstraps: which=0 -> c+t-1/8, which=1 -> -c+e-1/8     */
/* (HELM passes cond=input[2], then=input[0], else=input[1]: gates.rs:265).     */
/* ------------------------------------------------------------------------- */
void orc_gate_lincomb(int n, int op, int which, const u32 *in0, const u32 *in1,
                      const u32 *in2, u32 *out)
{
    for (int i = 0; i <= n; i++) {
        u32 cst_t = (i == n) ? PT_TRUE : 0, cst_f = (i == n) ? PT_FALSE : 0;
        u32 l = in0 ? in0[i] : 0, r = in1 ? in1[i] : 0, v;
        switch (op) {
        case ORC_AND:  v = l + r + cst_f; break;
        case ORC_OR:   v = l + r + cst_t; break;
        case ORC_NAND: v = (u32)0 - (l + r) + cst_t; break;
        case ORC_NOR:  v = (u32)0 - (l + r) + cst_f; break;
        case ORC_XOR:  v = 2u * (l + r + cst_t); break;
        case ORC_XNOR: v = 2u * ((u32)0 - (l + r + cst_t)); break;
        case ORC_MUX: {
            u32 c = in2[i];
            v = (which == 0) ? (c + l + cst_f) : ((u32)0 - c + r + cst_f);
            break;
        }
        default: v = 0; break;
        }
        out[i] = v;
    }
}

/* Gates that need no bootstrap (gates.rs:256,268,272-274): NOT negates the   */
/* ciphertext, BUF/DFF copy it, constants are trivial encryptions.             */
void orc_gate_linear_only(int n, int op, const u32 *in0, u32 *out)
{
    for (int i = 0; i <= n; i++) {
        switch (op) {
        case ORC_NOT: out[i] = (u32)0 - in0[i]; break;
        case ORC_BUF: case ORC_DFF: out[i] = in0[i]; break;
        case ORC_CONST_ONE:  out[i] = (i == n) ? PT_TRUE : 0; break;
        case ORC_CONST_ZERO: out[i] = (i == n) ? PT_FALSE : 0; break;
        default: out[i] = 0; break;
        }
    }
}

/* Modulus switch Z_2^32 -> Z_2N, round half up. */
u32 orc_modswitch(u32 x, int log2_2N)
{
    u32 r = (x >> (32 - log2_2N - 1)) + 1;
    return (r >> 1) & (((u32)1 << log2_2N) - 1);
}

/* Signed gadget decomposition (closest representable, balanced digits).
 * digits[0] is level 1 (weight 2^(32-logB)), digits[l-1] is level l. */
void orc_decompose(u32 x, int logB, int l, int32_t *digits)
{
    int rep = logB * l;
    u32 state = (rep == 32) ? x : ((x + ((u32)1 << (31 - rep))) >> (32 - rep));
    u32 mask = ((u32)1 << logB) - 1;
    for (int lev = l - 1; lev >= 0; lev--) {
        u32 d = state & mask;
        state >>= logB;
        u32 carry = (((d - 1u) | state) & d) >> (logB - 1);
        state += carry;
        digits[lev] = (int32_t)d - (int32_t)(carry << logB);
    }
}

/* (X^a * P)[j] for a in [0, 2N), negacyclic. */
static inline u32 rot_coeff(const u32 *P, int N, int j, int a)
{
    int idx = (j - a) & (2 * N - 1);
    return idx < N ? P[idx] : (u32)0 - P[idx - N];
}

/* ------------------------------------------------------------------------- */
/* Route 1: schoolbook external product, wrapping u32.                         */
/* bsk_i layout: [level j][row r][col c][N]; out[c] += sum digit(r,j) * row.    */
/* ------------------------------------------------------------------------- */
static void extprod_add_schoolbook(const orc_params *P, const u32 *bsk_i,
                                   const u32 *diff /* (k+1) x N */, u32 *acc)
{
    int N = P->N, k1 = P->k + 1, l = P->pbs_l;
    int32_t *dig = (int32_t *)malloc(sizeof(int32_t) * (size_t)k1 * l * N);
    int32_t tmp[64];
    for (int r = 0; r < k1; r++)
        for (int t = 0; t < N; t++) {
            orc_decompose(diff[r * N + t], P->pbs_logB, l, tmp);
            for (int j = 0; j < l; j++) dig[((size_t)r * l + j) * N + t] = tmp[j];
        }
    for (int j = 0; j < l; j++)
        for (int r = 0; r < k1; r++) {
            const int32_t *d = dig + ((size_t)r * l + j) * N;
            for (int c = 0; c < k1; c++) {
                const u32 *row = bsk_i + (((size_t)j * k1 + r) * k1 + c) * N;
                u32 *o = acc + (size_t)c * N;
                for (int a = 0; a < N; a++) {
                    u32 da = (u32)d[a];
                    if (!da) continue;
                    for (int b = 0; b < N - a; b++) o[a + b] += da * row[b];
                    for (int b = N - a; b < N; b++) o[a + b - N] -= da * row[b];
                }
            }
        }
    free(dig);
}

/* ------------------------------------------------------------------------- */
/* Route 2: Goldilocks NTT.                                                    */
/* ------------------------------------------------------------------------- */
#include "goldilocks.inc"

/* Bootstrapping key in the oracle's NTT domain. */
typedef struct {
    orc_params P;
    gl_tables *T;
    u64 *data; /* [n][l][k+1][k+1][N] */
} orc_bsk_ntt;

orc_bsk_ntt *orc_bsk_ntt_new(const orc_params *P, const u32 *bsk_std)
{
    orc_bsk_ntt *B = (orc_bsk_ntt *)malloc(sizeof(*B));
    B->P = *P; B->T = gl_tables_new(P->N);
    size_t polys = (size_t)P->n * P->pbs_l * (P->k + 1) * (P->k + 1);
    B->data = (u64 *)malloc(sizeof(u64) * polys * P->N);
    #pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)polys; q++) {
        u64 *dst = B->data + (size_t)q * P->N;
        const u32 *src = bsk_std + (size_t)q * P->N;
        for (int t = 0; t < P->N; t++) dst[t] = gl_from_i64((int64_t)(int32_t)src[t]);
        gl_ntt_fwd(B->T, dst);
        for (int t = 0; t < P->N; t++) dst[t] = gl_mul(dst[t], B->T->n_inv);
    }
    return B;
}
void orc_bsk_ntt_free(orc_bsk_ntt *B) { gl_tables_free(B->T); free(B->data); free(B); }

static void extprod_add_ntt(const orc_bsk_ntt *B, int i, const u32 *diff, u32 *acc, u64 *scratch)
{
    const orc_params *P = &B->P;
    int N = P->N, k1 = P->k + 1, l = P->pbs_l;
    u64 *f = scratch;                       /* k1*l polys */
    u64 *o = scratch + (size_t)k1 * l * N;  /* k1 polys   */
    int32_t tmp[64];
    for (int r = 0; r < k1; r++)
        for (int t = 0; t < N; t++) {
            orc_decompose(diff[r * N + t], P->pbs_logB, l, tmp);
            for (int j = 0; j < l; j++) f[((size_t)r * l + j) * N + t] = gl_from_i64(tmp[j]);
        }
    for (int q = 0; q < k1 * l; q++) gl_ntt_fwd(B->T, f + (size_t)q * N);
    memset(o, 0, sizeof(u64) * (size_t)k1 * N);
    const u64 *bi = B->data + (size_t)i * l * k1 * k1 * N;
    for (int j = 0; j < l; j++)
        for (int r = 0; r < k1; r++) {
            const u64 *fr = f + ((size_t)r * l + j) * N;
            for (int c = 0; c < k1; c++) {
                const u64 *row = bi + (((size_t)j * k1 + r) * k1 + c) * N;
                u64 *oc = o + (size_t)c * N;
                for (int t = 0; t < N; t++) oc[t] = gl_add(oc[t], gl_mul(fr[t], row[t]));
            }
        }
    for (int c = 0; c < k1; c++) {
        u64 *oc = o + (size_t)c * N;
        gl_ntt_inv(B->T, oc);
        for (int t = 0; t < N; t++) {
            /* centred lift then wrap mod 2^32 */
            u64 v = oc[t];
            u32 w = (v > GL_P / 2) ? (u32)0 - (u32)(GL_P - v) : (u32)v;
            acc[(size_t)c * N + t] += w;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Blind rotate + sample extract.  lwe: n+1 words (already the gate's linear   */
/* combination).  tv: N-word test polynomial (body), mask polys zero.          */
/* bsk_std may be NULL when bsk_ntt is given and vice versa.                   */
/* out_big: k*N+1 words under the GLWE key seen as an LWE key.                 */
/* ------------------------------------------------------------------------- */
void orc_bootstrap_noks(const orc_params *P, const u32 *bsk_std, const orc_bsk_ntt *bsk_ntt,
                        const u32 *lwe, const u32 *tv, u32 *out_big)
{
    int N = P->N, k1 = P->k + 1, n = P->n, l = P->pbs_l;
    int log2_2N = 1; while ((1 << log2_2N) < 2 * N) log2_2N++;
    u32 *acc = (u32 *)calloc((size_t)k1 * N, sizeof(u32));
    u32 *diff = (u32 *)malloc(sizeof(u32) * (size_t)k1 * N);
    u64 *scratch = bsk_ntt ? (u64 *)malloc(sizeof(u64) * (size_t)(k1 * l + k1) * N) : NULL;
    int bt = (int)orc_modswitch(lwe[n], log2_2N);
    /* acc = X^{-b~} * (0,...,0,tv) */
    for (int j = 0; j < N; j++) acc[(size_t)P->k * N + j] = rot_coeff(tv, N, j, (2 * N - bt) & (2 * N - 1));
    size_t bsk_stride = (size_t)l * k1 * k1 * N;
    for (int i = 0; i < n; i++) {
        int a = (int)orc_modswitch(lwe[i], log2_2N);
        if (a == 0) continue;
        for (int r = 0; r < k1; r++)
            for (int j = 0; j < N; j++)
                diff[r * N + j] = rot_coeff(acc + (size_t)r * N, N, j, a) - acc[(size_t)r * N + j];
        if (bsk_ntt) extprod_add_ntt(bsk_ntt, i, diff, acc, scratch);
        else extprod_add_schoolbook(P, bsk_std + (size_t)i * bsk_stride, diff, acc);
    }
    /* sample extract, coefficient 0 */
    for (int r = 0; r < P->k; r++) {
        const u32 *A = acc + (size_t)r * N;
        out_big[r * N] = A[0];
        for (int t = 1; t < N; t++) out_big[r * N + t] = (u32)0 - A[N - t];
    }
    out_big[P->k * N] = acc[(size_t)P->k * N];
    free(acc); free(diff); free(scratch);
}

/* ------------------------------------------------------------------------- */
/* Route 3: fp64-FMA NTT, one gate per SIMD lane (fp_route.inc).               */
/* ------------------------------------------------------------------------- */
typedef struct {
    int N, logN;
    double p, pinv, lim;          /* lim: largest |value| / p kept unreduced (0.95 * 2^53 / p) */
    double *psi_rev, *psi_inv_rev; /* bit-reversed powers of the 2N-th root, centred           */
    double n_inv;
} orc_fp_tables;

typedef struct {
    orc_params P;
    orc_fp_tables *T;
    double *data; /* [n][l][k+1][k+1][N], transform domain, centred, 1/N folded in */
} orc_bsk_fp;

/* |mulmod(a, w)| <= fp_mo(b) p for |a| <= b p, |w| <= p/2 (quotient estimate off by <= 1.5 b p / 2^53) */
static inline double fp_mo(const orc_fp_tables *T, double b) { return 0.5 + 1.5 * b * T->p / 9007199254740992.0 + 1e-9; }

static u64 mulmod_u64(u64 a, u64 b, u64 p) { return (u64)((u128)a * b % p); }
static u64 powmod_u64(u64 a, u64 e, u64 p) { u64 r = 1; while (e) { if (e & 1) r = mulmod_u64(r, a, p); a = mulmod_u64(a, a, p); e >>= 1; } return r; }
static double centred_d(u64 v, u64 p) { return v > p / 2 ? -(double)(p - v) : (double)v; }

static orc_fp_tables *fp_tables_new(int N, u64 p)
{
    orc_fp_tables *T = (orc_fp_tables *)malloc(sizeof(*T));
    int logN = 0; while ((1 << logN) < N) logN++;
    T->N = N; T->logN = logN; T->p = (double)p; T->pinv = 1.0 / (double)p;
    T->lim = 0.95 * 9007199254740992.0 / (double)p;
    u64 psi = 0;
    for (u64 g = 2; g < 64 && !psi; g++) { /* an element of order exactly 2N */
        u64 c = powmod_u64(g, (p - 1) / (2 * (u64)N), p);
        if (powmod_u64(c, (u64)N, p) == p - 1) psi = c;
    }
    u64 psi_inv = powmod_u64(psi, p - 2, p), a = 1, b = 1;
    T->psi_rev = (double *)malloc(sizeof(double) * N);
    T->psi_inv_rev = (double *)malloc(sizeof(double) * N);
    for (int i = 0; i < N; i++) {
        T->psi_rev[bitrev(i, logN)] = centred_d(a, p); T->psi_inv_rev[bitrev(i, logN)] = centred_d(b, p);
        a = mulmod_u64(a, psi, p); b = mulmod_u64(b, psi_inv, p);
    }
    T->n_inv = centred_d(powmod_u64((u64)N, p - 2, p), p);
    return T;
}

#define VI_ADD(a, b) ((a) + (b))
#define VI_SUB(a, b) ((a) - (b))
#define VI_AND(a, b) ((a) & (b))
#define VI_OR(a, b) ((a) | (b))
#define VI_SRL(a, c) ((a) >> (c))   /* unsigned lanes: logical */
#define VI_SLL(a, c) ((a) << (c))

/* ---- AVX2 + FMA, 4 gates per call ---- */
#pragma GCC push_options
#pragma GCC target("avx2,fma")
#define VL 4
#define FN(x) fp4_##x
typedef __m256d fp4_V;
typedef u32 fp4_VI __attribute__((vector_size(16)));
typedef int32_t fp4_VS __attribute__((vector_size(16)));
#define V fp4_V
#define VI fp4_VI
#define VSET1(c) _mm256_set1_pd(c)
#define VMUL(a, b) _mm256_mul_pd(a, b)
#define VADD(a, b) _mm256_add_pd(a, b)
#define VSUB(a, b) _mm256_sub_pd(a, b)
#define VFMSUB(a, b, c) _mm256_fmsub_pd(a, b, c)
#define VFNMADD(a, b, c) _mm256_fnmadd_pd(a, b, c)
#define VRND(a) _mm256_round_pd(a, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)
#define VI_SET1(c) ((fp4_VI){(u32)(c), (u32)(c), (u32)(c), (u32)(c)})
#define VI_TO_V(x) _mm256_cvtepi32_pd((__m128i)(x))
#define V_LOW32(x) ((fp4_VI)_mm256_castsi256_si128(_mm256_permutevar8x32_epi32(_mm256_castpd_si256(x), _mm256_setr_epi32(0, 2, 4, 6, 0, 2, 4, 6))))
#include "fp_route.inc"
#undef VL
#undef FN
#undef V
#undef VI
#undef VSET1
#undef VMUL
#undef VADD
#undef VSUB
#undef VFMSUB
#undef VFNMADD
#undef VRND
#undef VI_SET1
#undef VI_TO_V
#undef V_LOW32
#pragma GCC pop_options

/* ---- AVX-512F/DQ, 8 gates per call ---- */
#pragma GCC push_options
#pragma GCC target("avx512f,avx512dq,avx512vl,avx2,fma")
#define VL 8
#define FN(x) fp8_##x
typedef __m512d fp8_V;
typedef u32 fp8_VI __attribute__((vector_size(32)));
#define V fp8_V
#define VI fp8_VI
#define VSET1(c) _mm512_set1_pd(c)
#define VMUL(a, b) _mm512_mul_pd(a, b)
#define VADD(a, b) _mm512_add_pd(a, b)
#define VSUB(a, b) _mm512_sub_pd(a, b)
#define VFMSUB(a, b, c) _mm512_fmsub_pd(a, b, c)
#define VFNMADD(a, b, c) _mm512_fnmadd_pd(a, b, c)
#define VRND(a) _mm512_roundscale_pd(a, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)
#define VI_SET1(c) ((fp8_VI){(u32)(c), (u32)(c), (u32)(c), (u32)(c), (u32)(c), (u32)(c), (u32)(c), (u32)(c)})
#define VI_TO_V(x) _mm512_cvtepi32_pd((__m256i)(x))
#define V_LOW32(x) ((fp8_VI)_mm512_cvtepi64_epi32(_mm512_castpd_si512(x)))
#include "fp_route.inc"
#undef VL
#undef FN
#undef V
#undef VI
#undef VSET1
#undef VMUL
#undef VADD
#undef VSUB
#undef VFMSUB
#undef VFNMADD
#undef VRND
#undef VI_SET1
#undef VI_TO_V
#undef V_LOW32
#pragma GCC pop_options

static int fp_lanes(void)
{
    static int lanes = 0;
    if (!lanes) {
        const char *force = getenv("ORC_FP_LANES"); /* 4 forces the AVX2 build on an AVX-512 machine (tests) */
        __builtin_cpu_init();
        lanes = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && !(force && atoi(force) == 4)) ? 8 : 4;
        if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("fma")) lanes = -1;
    }
    return lanes;
}
const char *orc_ntt_route(void)
{
    return fp_lanes() == 8 ? "C restatement, exact fp64-FMA NTT over a 49/51-bit prime, AVX-512: 8 gates per call in SIMD lanes"
         : fp_lanes() == 4 ? "C restatement, exact fp64-FMA NTT over a 49/51-bit prime, AVX2: 4 gates per call in SIMD lanes"
                           : "scalar C restatement with a Goldilocks NTT";
}
int orc_fp_lanes(void) { return fp_lanes(); }

/* Bootstrapping key for route 3.  The prime: 0x24007A8500001 (49 bits; NOT the GPU's field for boolean sets)
 * when the exact products of the set stay below its half, else 0x6060002B00001 (51 bits); NULL when neither
 * holds them or the CPU lacks AVX2 + FMA. */
void orc_bsk_fp_free(orc_bsk_fp *B);
orc_bsk_fp *orc_bsk_fp_new(const orc_params *P, const u32 *bsk_std)
{
    if (fp_lanes() < 0) return NULL;
    const int N = P->N, k1 = P->k + 1, l = P->pbs_l;
    /* |exact coefficient| <= (k+1) l N (B/2) 2^31 */
    const u128 bound = (u128)k1 * l * N * ((u64)1 << (P->pbs_logB - 1)) * ((u64)1 << 31);
    const u64 p49 = 0x24007A8500001ull, p51 = 0x6060002B00001ull;
    const u64 p = bound < p49 / 2 ? p49 : bound < p51 / 2 ? p51 : 0;
    if (!p) return NULL;
    orc_bsk_fp *B = (orc_bsk_fp *)malloc(sizeof(*B));
    B->P = *P; B->T = fp_tables_new(N, p);
    const size_t polys = (size_t)P->n * l * k1 * k1;
    B->data = (double *)aligned_alloc(64, sizeof(double) * polys * N);
    const int lanes = fp_lanes();
    #pragma omp parallel for schedule(static)
    for (long q0 = 0; q0 < (long)polys; q0 += lanes) {
        const int cnt = (int)((long)polys - q0 < lanes ? (long)polys - q0 : lanes);
        if (lanes == 8) fp8_bsk_convert(B, bsk_std, (size_t)q0, cnt);
        else fp4_bsk_convert(B, bsk_std, (size_t)q0, cnt);
    }
    return B;
}
void orc_bsk_fp_free(orc_bsk_fp *B)
{
    if (!B) return;
    free(B->T->psi_rev); free(B->T->psi_inv_rev); free(B->T); free(B->data); free(B);
}
double orc_bsk_fp_prime(const orc_bsk_fp *B) { return B ? B->T->p : 0.0; }

/* `count` bootstraps (no keyswitch) by route 3; lwe / out: arrays of row pointers. */
void orc_bootstrap_noks_fp(const orc_bsk_fp *B, const u32 *const *lwe, const u32 *tv, u32 *const *out, int count)
{
    const int lanes = fp_lanes();
    for (int g = 0; g < count; g += lanes) {
        const int c = count - g < lanes ? count - g : lanes;
        if (lanes == 8) fp8_bootstrap(B, lwe + g, tv, out + g, c);
        else fp4_bootstrap(B, lwe + g, tv, out + g, c);
    }
}

/* Keyswitch big (k*N) -> small (n). ksk layout [k*N][ks_l][n+1]. */
void orc_keyswitch(const orc_params *P, const u32 *ksk, const u32 *in_big, u32 *out)
{
    int n = P->n, kN = P->k * P->N, l = P->ks_l;
    int32_t dig[64];
    memset(out, 0, sizeof(u32) * (size_t)(n + 1));
    out[n] = in_big[kN];
    for (int t = 0; t < kN; t++) {
        orc_decompose(in_big[t], P->ks_logB, l, dig);
        for (int j = 0; j < l; j++) {
            u32 d = (u32)dig[j];
            if (!d) continue;
            const u32 *row = ksk + ((size_t)t * l + j) * (n + 1);
            for (int c = 0; c <= n; c++) out[c] -= d * row[c];
        }
    }
}

/* One full gate, bootstrapped or not.  Scratch allocated internally. */
void orc_gate(const orc_params *P, const u32 *bsk_std, const orc_bsk_ntt *bsk_ntt, const u32 *ksk,
              int op, const u32 *in0, const u32 *in1, const u32 *in2, u32 *out)
{
    int n = P->n, N = P->N, kN = P->k * N;
    if (op == ORC_NOT || op == ORC_BUF || op == ORC_DFF || op == ORC_CONST_ONE || op == ORC_CONST_ZERO) {
        orc_gate_linear_only(n, op, in0, out);
        return;
    }
    u32 *lin = (u32 *)malloc(sizeof(u32) * (size_t)(n + 1));
    u32 *big = (u32 *)malloc(sizeof(u32) * (size_t)(kN + 1));
    u32 *tv = (u32 *)malloc(sizeof(u32) * (size_t)N);
    for (int j = 0; j < N; j++) tv[j] = PT_TRUE;
    orc_gate_lincomb(n, op, 0, in0, in1, in2, lin);
    orc_bootstrap_noks(P, bsk_std, bsk_ntt, lin, tv, big);
    if (op == ORC_MUX) {
        u32 *big2 = (u32 *)malloc(sizeof(u32) * (size_t)(kN + 1));
        orc_gate_lincomb(n, op, 1, in0, in1, in2, lin);
        orc_bootstrap_noks(P, bsk_std, bsk_ntt, lin, tv, big2);
        for (int t = 0; t <= kN; t++) big[t] += big2[t];
        big[kN] += PT_TRUE;
        free(big2);
    }
    orc_keyswitch(P, ksk, big, out);
    free(lin); free(big); free(tv);
}

/* A netlist level: `count` independent gates over a wire table
 * (rows of n+1 words), parallel over gates like rayon's par_iter_mut
 * (reference src/circuit.rs:531).  Index -1 = unused operand. */
void orc_eval_level(const orc_params *P, const u32 *bsk_std, const orc_bsk_ntt *bsk_ntt, const u32 *ksk,
                    u32 *wires, const int32_t *opcode, const int32_t *in0, const int32_t *in1,
                    const int32_t *in2, const int32_t *outw, int count, int nthreads)
{
    size_t row = (size_t)P->n + 1;
    u32 *tmp = (u32 *)malloc(sizeof(u32) * row * (size_t)count);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    #pragma omp parallel for schedule(dynamic, 1)
    for (int g = 0; g < count; g++) {
        const u32 *a = in0[g] >= 0 ? wires + row * (size_t)in0[g] : NULL;
        const u32 *b = in1[g] >= 0 ? wires + row * (size_t)in1[g] : NULL;
        const u32 *c = in2[g] >= 0 ? wires + row * (size_t)in2[g] : NULL;
        orc_gate(P, bsk_std, bsk_ntt, ksk, opcode[g], a, b, c, tmp + row * (size_t)g);
    }
    for (int g = 0; g < count; g++) memcpy(wires + row * (size_t)outw[g], tmp + row * (size_t)g, sizeof(u32) * row);
    free(tmp);
}

/* The same level by route 3: the bootstraps of the level (a MUX is two) in groups of one SIMD vector of
 * gates, OpenMP over the groups; then the keyswitch per gate.  Bit-identical to orc_eval_level. */
void orc_eval_level_fp(const orc_params *P, const orc_bsk_fp *B, const u32 *ksk, u32 *wires, const int32_t *opcode,
                       const int32_t *in0, const int32_t *in1, const int32_t *in2, const int32_t *outw, int count,
                       int nthreads)
{
    const size_t row = (size_t)P->n + 1, brow = (size_t)P->k * P->N + 1;
    const int lanes = fp_lanes();
    int n_boot = 0;
    for (int g = 0; g < count; g++)
        n_boot += opcode[g] == ORC_MUX ? 2 : (opcode[g] == ORC_NOT || opcode[g] == ORC_BUF || opcode[g] == ORC_DFF ||
                                              opcode[g] == ORC_CONST_ONE || opcode[g] == ORC_CONST_ZERO) ? 0 : 1;
    u32 *tmp = (u32 *)malloc(sizeof(u32) * row * (size_t)count);
    u32 *lin = (u32 *)malloc(sizeof(u32) * row * (size_t)(n_boot + 1));
    u32 *big = (u32 *)malloc(sizeof(u32) * brow * (size_t)(n_boot + 1));
    const u32 **lp = (const u32 **)malloc(sizeof(u32 *) * (size_t)(n_boot + 1));
    u32 **bp = (u32 **)malloc(sizeof(u32 *) * (size_t)(n_boot + 1));
    int *first = (int *)malloc(sizeof(int) * (size_t)(count + 1));
    u32 *tv = (u32 *)malloc(sizeof(u32) * (size_t)P->N);
    for (int j = 0; j < P->N; j++) tv[j] = PT_TRUE;
    int nb = 0;
    for (int g = 0; g < count; g++) {
        const u32 *a = in0[g] >= 0 ? wires + row * (size_t)in0[g] : NULL;
        const u32 *b = in1[g] >= 0 ? wires + row * (size_t)in1[g] : NULL;
        const u32 *c = in2[g] >= 0 ? wires + row * (size_t)in2[g] : NULL;
        first[g] = nb;
        const int op = opcode[g];
        if (op == ORC_NOT || op == ORC_BUF || op == ORC_DFF || op == ORC_CONST_ONE || op == ORC_CONST_ZERO) {
            orc_gate_linear_only(P->n, op, a, tmp + row * (size_t)g);
            continue;
        }
        for (int which = 0; which < (op == ORC_MUX ? 2 : 1); which++, nb++) {
            orc_gate_lincomb(P->n, op, which, a, b, c, lin + row * (size_t)nb);
            lp[nb] = lin + row * (size_t)nb;
            bp[nb] = big + brow * (size_t)nb;
        }
    }
    first[count] = nb;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    #pragma omp parallel for schedule(dynamic, 1)
    for (int g = 0; g < nb; g += lanes)
        orc_bootstrap_noks_fp(B, lp + g, tv, bp + g, nb - g < lanes ? nb - g : lanes);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int g = 0; g < count; g++) {
        if (first[g + 1] == first[g]) continue;
        u32 *bg = big + brow * (size_t)first[g];
        if (opcode[g] == ORC_MUX) {
            const u32 *b2 = bg + brow;
            for (size_t t = 0; t < brow; t++) bg[t] += b2[t];
            bg[brow - 1] += PT_TRUE;
        }
        orc_keyswitch(P, ksk, bg, tmp + row * (size_t)g);
    }
    for (int g = 0; g < count; g++) memcpy(wires + row * (size_t)outw[g], tmp + row * (size_t)g, sizeof(u32) * row);
    free(tmp); free(lin); free(big); free(lp); free(bp); free(first); free(tv);
}

/* phase = b - <a, s>; decrypt: phase < 2^31 => true (reference src/circuit.rs:948) */
u32 orc_phase(int n, const u32 *sk_bits, const u32 *ct)
{
    u32 ph = ct[n];
    for (int i = 0; i < n; i++) if (sk_bits[i]) ph -= ct[i];
    return ph;
}
int orc_decrypt_bool(int n, const u32 *sk_bits, const u32 *ct) { return orc_phase(n, sk_bits, ct) < ((u32)1 << 31); }

/* Exposed for unit tests: one external-product accumulate by either route. */
void orc_extprod_add(const orc_params *P, const u32 *bsk_i_std, const orc_bsk_ntt *bsk_ntt, int i,
                     const u32 *diff, u32 *acc)
{
    if (bsk_ntt) {
        u64 *scratch = (u64 *)malloc(sizeof(u64) * (size_t)((P->k + 1) * P->pbs_l + P->k + 1) * P->N);
        extprod_add_ntt(bsk_ntt, i, diff, acc, scratch);
        free(scratch);
    } else extprod_add_schoolbook(P, bsk_i_std, diff, acc);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
