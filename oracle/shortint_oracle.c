/*
 * shortint_oracle.c — CPU restatement of the HELM LUT-mode / arithmetic-mode hot path
 * (64-bit torus, KS_PBS order).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as tfhe_oracle.c): nothing under helm_amd/ may
 * include, link or call this file.
 *
 * What it restates.  HELM's LUT gate (reference src/gates.rs:754-785) calls
 * tfhe::shortint::ServerKey::{smart_scalar_left_shift, add, generate_lookup_table,
 * apply_lookup_table, smart_evaluate_bivariate_function, smart_neg}; the arithmetic lives
 * in the third-party crate `tfhe = 0.4.1` (reference Cargo.toml:18), absent from
 * /root/reference and unbuildable here.  This file restates the published algorithm:
 * apply_lookup_table = LWE keyswitch (big -> small key) then programmable bootstrap
 * (modulus switch, blind rotate with the CMUX / external product over the GGSW
 * bootstrapping key, sample extract), with the shortint encoding delta = 2^63/(msg*carry)
 * (the one citable line: src/gates.rs:851) and generate_lookup_table's box layout.
 * Gate semantics follow src/gates.rs:746-785 (index convention: first input = MSB,
 * src/gates.rs:159-167).
 *
 * PARITY STATUS: "parity unpinned" at ciphertext level, as for tfhe_oracle.c; pinned at
 * the decrypted level by the reference's own LUT test (tests/circuit_test.rs:308-310:
 * every wire of the 8-bit LUT adder equals the plaintext evaluation).
 *
 * Exactness: negacyclic products two ways that must agree bit for bit
 * (tests/test_oracle_shortint.py): schoolbook convolution in wrapping u64 arithmetic
 * (a ring homomorphism Z -> Z/2^64, obviously correct, O(N^2)), and - round 5, the route
 * that makes whole levels at the full parameter sets checkable - a Goldilocks NTT on the
 * key split into parts of 32, 16 or 8 bits so that every exact integer sum
 * (digits x part x N x (k+1) l) stays below 2^63, the parts recombined mod 2^64 (the same
 * construction as oracle/wopbs_oracle.c).  The HIP path computes the same exact integers
 * with two fp64 NTT fields + CRT, so ciphertexts must agree bit for bit.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#include "goldilocks.inc"

typedef struct {
    int32_t n, k, N;
    int32_t pbs_l, pbs_logB;
    int32_t ks_l, ks_logB;
    int32_t message_modulus, carry_modulus;
    int32_t grouping_factor; /* 0 / 1: classical blind rotation; 2, 3: multi-bit (helm_shortint.h) */
} orc64_params;

u64 orc64_delta(const orc64_params *P) { return ((u64)1 << 63) / (u64)(P->message_modulus * P->carry_modulus); }

u64 orc64_modswitch(u64 x, int log2_2N)
{
    u64 r = (x >> (64 - log2_2N - 1)) + 1;
    return (r >> 1) & (((u64)1 << log2_2N) - 1);
}

/* signed gadget decomposition, digits[0] = most significant level */
void orc64_decompose(u64 x, int logB, int l, int64_t *digits)
{
    int rep = logB * l;
    u64 state = (rep >= 64) ? x : ((x + ((u64)1 << (63 - rep))) >> (64 - rep));
    u64 mask = ((u64)1 << logB) - 1;
    for (int lev = l - 1; lev >= 0; lev--) {
        u64 d = state & mask;
        state >>= logB;
        u64 carry = (((d - 1) | state) & d) >> (logB - 1);
        state += carry;
        digits[lev] = (int64_t)d - (int64_t)(carry << logB);
    }
}

/* ServerKey::generate_lookup_table(f): f given as its value table over [0, msg*carry) */
void orc64_make_lut(const orc64_params *P, const u64 *f_values, u64 *tv)
{
    int N = P->N, t = P->message_modulus * P->carry_modulus, box = N / t, half = box / 2;
    u64 delta = orc64_delta(P);
    u64 *acc = (u64 *)malloc(sizeof(u64) * (size_t)N);
    for (int v = 0; v < t; v++)
        for (int j = 0; j < box; j++) acc[v * box + j] = f_values[v] * delta;
    for (int j = 0; j < half; j++) acc[j] = (u64)0 - acc[j];
    for (int j = 0; j < N; j++) tv[j] = acc[(j + half) % N];
    free(acc);
}

static inline u64 rot_coeff64(const u64 *P, int N, int j, int a)
{
    int idx = (j - a) & (2 * N - 1);
    return idx < N ? P[idx] : (u64)0 - P[idx - N];
}

/* acc += BSK_i (x) diff, schoolbook; bsk_i layout [level j][row r][col c][N] */
static void extprod_add64(const orc64_params *P, const u64 *bsk_i, const u64 *diff, u64 *acc)
{
    int N = P->N, k1 = P->k + 1, l = P->pbs_l;
    int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * (size_t)k1 * l * N);
    int64_t tmp[64];
    for (int r = 0; r < k1; r++)
        for (int t = 0; t < N; t++) {
            orc64_decompose(diff[r * N + t], P->pbs_logB, l, tmp);
            for (int j = 0; j < l; j++) dig[((size_t)r * l + j) * N + t] = tmp[j];
        }
    for (int j = 0; j < l; j++)
        for (int r = 0; r < k1; r++) {
            const int64_t *d = dig + ((size_t)r * l + j) * N;
            for (int c = 0; c < k1; c++) {
                const u64 *row = bsk_i + (((size_t)j * k1 + r) * k1 + c) * N;
                u64 *o = acc + (size_t)c * N;
                for (int a = 0; a < N; a++) {
                    u64 da = (u64)d[a];
                    if (!da) continue;
                    for (int b = 0; b < N - a; b++) o[a + b] += da * row[b];
                    for (int b = N - a; b < N; b++) o[a + b - N] -= da * row[b];
                }
            }
        }
    free(dig);
}

/* programmable bootstrap: small LWE (n+1) -> big LWE (k*N+1) with look-up table tv */
void orc64_bootstrap(const orc64_params *P, const u64 *bsk, const u64 *lwe, const u64 *tv, u64 *out_big)
{
    int N = P->N, k1 = P->k + 1, n = P->n, l = P->pbs_l;
    int log2_2N = 1; while ((1 << log2_2N) < 2 * N) log2_2N++;
    u64 *acc = (u64 *)calloc((size_t)k1 * N, sizeof(u64));
    u64 *diff = (u64 *)malloc(sizeof(u64) * (size_t)k1 * N);
    int bt = (int)orc64_modswitch(lwe[n], log2_2N);
    for (int j = 0; j < N; j++) acc[(size_t)P->k * N + j] = rot_coeff64(tv, N, j, (2 * N - bt) & (2 * N - 1));
    size_t stride = (size_t)l * k1 * k1 * N;
    int g = P->grouping_factor;
    if (g > 1) {
        /* Multi-bit blind rotation (tfhe MultiBitPBS, the parameter set of reference
         * src/bin/helm.rs:83; algorithm as published): per group of g mask words
         *   G = sum_{S subset of group} X^(sum_{i in S} a~_i) * GGSW_S,   acc <- G (x) acc
         * where GGSW_S encrypts the indicator that exactly the key bits of S are set, so G is a GGSW
         * of X^(sum_i a~_i s_i).  bsk layout [n/g][2^g][level][row][col][N]. */
        int subsets = 1 << g;
        u64 *G = (u64 *)malloc(sizeof(u64) * stride);
        for (int t = 0; t < n / g; t++) {
            memset(G, 0, sizeof(u64) * stride);
            for (int S = 0; S < subsets; S++) {
                int e = 0;
                for (int i = 0; i < g; i++)
                    if ((S >> i) & 1) e += (int)orc64_modswitch(lwe[t * g + i], log2_2N);
                e &= 2 * N - 1;
                const u64 *src = bsk + ((size_t)t * subsets + S) * stride;
                for (size_t q = 0; q < stride / N; q++)
                    for (int j = 0; j < N; j++) G[q * N + j] += rot_coeff64(src + q * N, N, j, e);
            }
            memcpy(diff, acc, sizeof(u64) * (size_t)k1 * N);
            memset(acc, 0, sizeof(u64) * (size_t)k1 * N);
            extprod_add64(P, G, diff, acc);
        }
        free(G);
    } else
    for (int i = 0; i < n; i++) {
        int a = (int)orc64_modswitch(lwe[i], log2_2N);
        if (a == 0) continue;
        for (int r = 0; r < k1; r++)
            for (int j = 0; j < N; j++)
                diff[r * N + j] = rot_coeff64(acc + (size_t)r * N, N, j, a) - acc[(size_t)r * N + j];
        extprod_add64(P, bsk + (size_t)i * stride, diff, acc);
    }
    for (int r = 0; r < P->k; r++) {
        const u64 *A = acc + (size_t)r * N;
        out_big[r * N] = A[0];
        for (int t = 1; t < N; t++) out_big[r * N + t] = (u64)0 - A[N - t];
    }
    out_big[P->k * N] = acc[(size_t)P->k * N];
    free(acc); free(diff);
}

/* ------------------------------------------------------------------------- */
/* The second exact route: the bootstrapping key behind a handle, Goldilocks NTT */
/* ------------------------------------------------------------------------- */
typedef struct {
    orc64_params P;
    const u64 *std;       /* the standard-domain key, borrowed */
    int use_ntt, parts, part_bits;
    gl_tables *T;
    u64 *ntt;             /* classical rotation: [n][level][row][col][part][N], transform domain, 1/N folded in */
} orc64_key;

static int log2c(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

/* one GGSW [level][row][col][N] (standard domain, mod 2^64) -> its parts in the transform domain */
static void ggsw_to_ntt(const orc64_key *K, const u64 *ggsw_std, u64 *dst)
{
    int N = K->P.N, k1 = K->P.k + 1;
    size_t polys = (size_t)K->P.pbs_l * k1 * k1;
    u64 pmask = ((u64)1 << K->part_bits) - 1;
    for (size_t q = 0; q < polys; q++)
        for (int p = 0; p < K->parts; p++) {
            u64 *d = dst + (q * K->parts + p) * N;
            for (int t = 0; t < N; t++) d[t] = (ggsw_std[q * N + t] >> (p * K->part_bits)) & pmask;
            gl_ntt_fwd(K->T, d);
            for (int t = 0; t < N; t++) d[t] = gl_mul(d[t], K->T->n_inv);
        }
}

orc64_key *orc64_key_new(const orc64_params *P, const u64 *bsk_std, int use_ntt)
{
    orc64_key *K = (orc64_key *)calloc(1, sizeof(*K));
    K->P = *P;
    K->std = bsk_std;
    K->use_ntt = use_ntt;
    if (!use_ntt) return K;
    int k1 = P->k + 1;
    /* |sum| <= (k+1) l N (B/2) 2^part_bits must stay below 2^63 */
    int budget = 63 - (log2c(k1 * P->pbs_l) + log2c(P->N) + P->pbs_logB - 1);
    K->part_bits = budget >= 32 ? 32 : budget >= 16 ? 16 : 8;
    K->parts = 64 / K->part_bits;
    K->T = gl_tables_new(P->N);
    if (P->grouping_factor > 1) return K; /* multi-bit: the key of a step is a sum that depends on the ciphertext: transformed per step */
    size_t stride = (size_t)P->pbs_l * k1 * k1 * P->N;
    K->ntt = (u64 *)malloc(sizeof(u64) * stride * K->parts * (size_t)P->n);
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < P->n; i++) ggsw_to_ntt(K, bsk_std + (size_t)i * stride, K->ntt + (size_t)i * stride * K->parts);
    return K;
}
void orc64_key_free(orc64_key *K)
{
    if (!K) return;
    if (K->T) gl_tables_free(K->T);
    free(K->ntt);
    free(K);
}
int orc64_key_part_bits(const orc64_key *K) { return K->use_ntt ? K->part_bits : 0; }

/* acc += GGSW (x) diff with the GGSW given as transform-domain parts [level][row][col][part][N] */
static void extprod_add64_ntt(const orc64_key *K, const u64 *gi, const u64 *diff, u64 *acc)
{
    const orc64_params *P = &K->P;
    int N = P->N, k1 = P->k + 1, l = P->pbs_l;
    u64 *f = (u64 *)malloc(sizeof(u64) * (size_t)k1 * l * N);
    int64_t tmp[64];
    for (int r = 0; r < k1; r++)
        for (int t = 0; t < N; t++) {
            orc64_decompose(diff[r * N + t], P->pbs_logB, l, tmp);
            for (int j = 0; j < l; j++) f[((size_t)r * l + j) * N + t] = gl_from_i64(tmp[j]);
        }
    for (int q = 0; q < k1 * l; q++) gl_ntt_fwd(K->T, f + (size_t)q * N);
    u64 *o = (u64 *)malloc(sizeof(u64) * (size_t)N);
    for (int c = 0; c < k1; c++)
        for (int p = 0; p < K->parts; p++) {
            memset(o, 0, sizeof(u64) * (size_t)N);
            for (int j = 0; j < l; j++)
                for (int r = 0; r < k1; r++) {
                    const u64 *fr = f + ((size_t)r * l + j) * N;
                    const u64 *row = gi + (((((size_t)j * k1 + r) * k1 + c) * K->parts) + p) * N;
                    for (int t = 0; t < N; t++) o[t] = gl_add(o[t], gl_mul(fr[t], row[t]));
                }
            gl_ntt_inv(K->T, o);
            u64 *A = acc + (size_t)c * N;
            for (int t = 0; t < N; t++) {
                /* centred lift: the exact integer is below 2^63 in magnitude */
                u64 v = o[t] > GL_P / 2 ? o[t] - GL_P : o[t]; /* two's complement of the negative representative */
                A[t] += v << (p * K->part_bits);
            }
        }
    free(o);
    free(f);
}

/* orc64_bootstrap through the handle: schoolbook when the key was made with use_ntt = 0 (then identical to
 * orc64_bootstrap), the NTT route otherwise.  Same steps, same order, same integers. */
void orc64_bootstrap_k(const orc64_key *K, const u64 *lwe, const u64 *tv, u64 *out_big)
{
    const orc64_params *P = &K->P;
    if (!K->use_ntt) {
        orc64_bootstrap(P, K->std, lwe, tv, out_big);
        return;
    }
    int N = P->N, k1 = P->k + 1, n = P->n, l = P->pbs_l;
    int log2_2N = 1; while ((1 << log2_2N) < 2 * N) log2_2N++;
    u64 *acc = (u64 *)calloc((size_t)k1 * N, sizeof(u64));
    u64 *diff = (u64 *)malloc(sizeof(u64) * (size_t)k1 * N);
    int bt = (int)orc64_modswitch(lwe[n], log2_2N);
    for (int j = 0; j < N; j++) acc[(size_t)P->k * N + j] = rot_coeff64(tv, N, j, (2 * N - bt) & (2 * N - 1));
    size_t stride = (size_t)l * k1 * k1 * N;
    int g = P->grouping_factor;
    if (g > 1) {
        int subsets = 1 << g;
        u64 *G = (u64 *)malloc(sizeof(u64) * stride);
        u64 *Gn = (u64 *)malloc(sizeof(u64) * stride * K->parts);
        for (int t = 0; t < n / g; t++) {
            memset(G, 0, sizeof(u64) * stride);
            for (int S = 0; S < subsets; S++) {
                int e = 0;
                for (int i = 0; i < g; i++)
                    if ((S >> i) & 1) e += (int)orc64_modswitch(lwe[t * g + i], log2_2N);
                e &= 2 * N - 1;
                const u64 *src = K->std + ((size_t)t * subsets + S) * stride;
                for (size_t q = 0; q < stride / N; q++)
                    for (int j = 0; j < N; j++) G[q * N + j] += rot_coeff64(src + q * N, N, j, e);
            }
            ggsw_to_ntt(K, G, Gn);
            memcpy(diff, acc, sizeof(u64) * (size_t)k1 * N);
            memset(acc, 0, sizeof(u64) * (size_t)k1 * N);
            extprod_add64_ntt(K, Gn, diff, acc);
        }
        free(G); free(Gn);
    } else
    for (int i = 0; i < n; i++) {
        int a = (int)orc64_modswitch(lwe[i], log2_2N);
        if (a == 0) continue;
        for (int r = 0; r < k1; r++)
            for (int j = 0; j < N; j++)
                diff[r * N + j] = rot_coeff64(acc + (size_t)r * N, N, j, a) - acc[(size_t)r * N + j];
        extprod_add64_ntt(K, K->ntt + (size_t)i * stride * K->parts, diff, acc);
    }
    for (int r = 0; r < P->k; r++) {
        const u64 *A = acc + (size_t)r * N;
        out_big[r * N] = A[0];
        for (int t = 1; t < N; t++) out_big[r * N + t] = (u64)0 - A[N - t];
    }
    out_big[P->k * N] = acc[(size_t)P->k * N];
    free(acc); free(diff);
}

/* keyswitch big (k*N) -> small (n); ksk layout [k*N][ks_l][n+1] */
void orc64_keyswitch(const orc64_params *P, const u64 *ksk, const u64 *in_big, u64 *out)
{
    int n = P->n, kN = P->k * P->N, l = P->ks_l;
    int64_t dig[64];
    memset(out, 0, sizeof(u64) * (size_t)(n + 1));
    out[n] = in_big[kN];
    for (int t = 0; t < kN; t++) {
        orc64_decompose(in_big[t], P->ks_logB, l, dig);
        for (int j = 0; j < l; j++) {
            u64 d = (u64)dig[j];
            if (!d) continue;
            const u64 *row = ksk + ((size_t)t * l + j) * (n + 1);
            for (int c = 0; c <= n; c++) out[c] -= d * row[c];
        }
    }
}

/* ServerKey::apply_lookup_table (KS_PBS): big -> big */
void orc64_apply_lut(const orc64_params *P, const u64 *bsk, const u64 *ksk, const u64 *in_big, const u64 *tv,
                     u64 *out_big)
{
    u64 *small = (u64 *)malloc(sizeof(u64) * (size_t)(P->n + 1));
    orc64_keyswitch(P, ksk, in_big, small);
    orc64_bootstrap(P, bsk, small, tv, out_big);
    free(small);
}

void orc64_apply_lut_k(const orc64_key *K, const u64 *ksk, const u64 *in_big, const u64 *tv, u64 *out_big)
{
    u64 *small = (u64 *)malloc(sizeof(u64) * (size_t)(K->P.n + 1));
    orc64_keyswitch(&K->P, ksk, in_big, small);
    orc64_bootstrap_k(K, small, tv, out_big);
    free(small);
}

/* out = sum coef[t] * in[t] + const_add * delta  (rows of dim+1 words; in[t] may be NULL) */
void orc64_lincomb(int dim, u64 delta, int terms, const u64 *const *in, const int64_t *coef, int64_t const_add, u64 *out)
{
    u64 *tmp = (u64 *)calloc((size_t)dim + 1, sizeof(u64));
    for (int t = 0; t < terms; t++) {
        if (!in[t]) continue;
        for (int i = 0; i <= dim; i++) tmp[i] += (u64)coef[t] * in[t][i];
    }
    tmp[dim] += (u64)const_add * delta;
    memcpy(out, tmp, sizeof(u64) * ((size_t)dim + 1));
    free(tmp);
}

/* gates::lut() (reference src/gates.rs:754-785) on big-LWE rows.  table: bit i = entry i. */
void orc64_lut_gate(const orc64_params *P, const u64 *bsk, const u64 *ksk, int arity, const u64 *const *in,
                    u64 table, u64 *out)
{
    int dim = P->k * P->N, t = P->message_modulus * P->carry_modulus;
    u64 delta = orc64_delta(P);
    int64_t coef[16];
    if (arity <= 1) { /* copy, or arithmetic negation when the table is not all zero (:765-770) */
        coef[0] = (arity == 1 && table != 0) ? -1 : 1;
        orc64_lincomb(dim, delta, 1, in, coef, 0, out);
        return;
    }
    for (int q = 0; q < arity; q++) coef[q] = (int64_t)1 << (arity - 1 - q); /* :773-778 */
    u64 *packed = (u64 *)malloc(sizeof(u64) * ((size_t)dim + 1));
    orc64_lincomb(dim, delta, arity, in, coef, 0, packed);
    u64 *f = (u64 *)malloc(sizeof(u64) * (size_t)t);
    for (int v = 0; v < t; v++) {
        int idx = arity == 2 ? (((v >> 1) & 1) * 2 + (v & 1)) : (v & ((1 << arity) - 1)); /* :746-752 */
        f[v] = (table >> idx) & 1;
    }
    u64 *tv = (u64 *)malloc(sizeof(u64) * (size_t)P->N);
    orc64_make_lut(P, f, tv);
    orc64_apply_lut(P, bsk, ksk, packed, tv, out);
    free(packed); free(f); free(tv);
}

/* orc64_lut_gate through the key handle (either route) */
void orc64_lut_gate_k(const orc64_key *K, const u64 *ksk, int arity, const u64 *const *in, u64 table, u64 *out)
{
    const orc64_params *P = &K->P;
    if (arity <= 1) {
        orc64_lut_gate(P, K->std, ksk, arity, in, table, out);
        return;
    }
    int dim = P->k * P->N, t = P->message_modulus * P->carry_modulus;
    u64 delta = orc64_delta(P);
    int64_t coef[16];
    for (int q = 0; q < arity; q++) coef[q] = (int64_t)1 << (arity - 1 - q); /* :773-778 */
    u64 *packed = (u64 *)malloc(sizeof(u64) * ((size_t)dim + 1));
    orc64_lincomb(dim, delta, arity, in, coef, 0, packed);
    u64 *f = (u64 *)malloc(sizeof(u64) * (size_t)t);
    for (int v = 0; v < t; v++) {
        int idx = arity == 2 ? (((v >> 1) & 1) * 2 + (v & 1)) : (v & ((1 << arity) - 1)); /* :746-752 */
        f[v] = (table >> idx) & 1;
    }
    u64 *tv = (u64 *)malloc(sizeof(u64) * (size_t)P->N);
    orc64_make_lut(P, f, tv);
    orc64_apply_lut_k(K, ksk, packed, tv, out);
    free(packed); free(f); free(tv);
}

/* the gates [first, first + count) of a level, results into out_rows (count rows): lets a test check chosen rows of a
 * level the GPU evaluated as a whole without paying for the others */
void orc64_eval_lut_rows_k(const orc64_key *K, const u64 *ksk, const u64 *wires, const int32_t *arity, const int32_t *in_idx,
                           int max_in, const u64 *table, const int32_t *gates, int count, u64 *out_rows)
{
    size_t row = (size_t)K->P.k * K->P.N + 1;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < count; q++) {
        int g = gates[q];
        const u64 *in[16];
        int ar = arity[g] < 1 ? 1 : arity[g];
        for (int x = 0; x < ar; x++) in[x] = wires + row * (size_t)in_idx[(size_t)g * max_in + x];
        orc64_lut_gate_k(K, ksk, arity[g], in, table[g], out_rows + row * (size_t)q);
    }
}

/* a batch of independent look-ups (helm_si_apply_luts): out_rows[q] = apply_lut(in_rows[q], luts[lut_index[q]]) */
void orc64_apply_luts_k(const orc64_key *K, const u64 *ksk, const u64 *in_rows, const u64 *luts, const int32_t *lut_index,
                        int count, u64 *out_rows)
{
    size_t row = (size_t)K->P.k * K->P.N + 1;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < count; q++)
        orc64_apply_lut_k(K, ksk, in_rows + row * (size_t)q, luts + (size_t)lut_index[q] * K->P.N, out_rows + row * (size_t)q);
}

/* a level of independent LUT gates (rayon par_iter_mut, reference src/circuit.rs:1055) */
void orc64_eval_lut_level(const orc64_params *P, const u64 *bsk, const u64 *ksk, u64 *wires, const int32_t *arity,
                          const int32_t *in_idx, int max_in, const u64 *table, const int32_t *out_idx, int count)
{
    size_t row = (size_t)P->k * P->N + 1;
    u64 *tmp = (u64 *)malloc(sizeof(u64) * row * (size_t)count);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int g = 0; g < count; g++) {
        const u64 *in[16];
        int ar = arity[g] < 1 ? 1 : arity[g];
        for (int q = 0; q < ar; q++) in[q] = wires + row * (size_t)in_idx[(size_t)g * max_in + q];
        orc64_lut_gate(P, bsk, ksk, arity[g], in, table[g], tmp + row * (size_t)g);
    }
    for (int g = 0; g < count; g++) memcpy(wires + row * (size_t)out_idx[g], tmp + row * (size_t)g, sizeof(u64) * row);
    free(tmp);
}

u64 orc64_phase(int dim, const u64 *sk_bits, const u64 *ct)
{
    u64 ph = ct[dim];
    for (int i = 0; i < dim; i++) if (sk_bits[i]) ph -= ct[i];
    return ph;
}
/* message and carry: round(phase / delta) mod (msg*carry) */
u64 orc64_decrypt(const orc64_params *P, const u64 *glwe_sk_bits, const u64 *ct)
{
    u64 delta = orc64_delta(P), t = (u64)(P->message_modulus * P->carry_modulus);
    u64 ph = orc64_phase(P->k * P->N, glwe_sk_bits, ct);
    return ((ph + delta / 2) / delta) % t;
}
