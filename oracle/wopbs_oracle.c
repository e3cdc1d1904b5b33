/*
 * wopbs_oracle.c — CPU restatement of the WoP-PBS wide-LUT path (bit extraction, circuit bootstrap,
 * vertical packing) on the 64-bit torus.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as tfhe_oracle.c): nothing under helm_amd/ may include, link or
 * call this file.
 *
 * What it restates.  HELM's Gate::evaluate_encrypted_high_precision_lut (reference src/gates.rs:721-742)
 * calls high_precision_lut (:787-815): input blocks -> radix ciphertext (first input = most significant
 * block, :795-799), WopbsKey::keyswitch_to_wopbs_params (:802), the table of
 * generate_high_precision_lut_radix_helm (:817-864), WopbsKey::wopbs (:808), keyswitch_to_pbs_params
 * (:811), block 0 returned (:814).  The arithmetic lives in the third-party crate tfhe = 0.4.1
 * (reference Cargo.toml:18), absent from /root/reference and unbuildable here; this file restates the
 * published algorithm (Bergerat et al., "Parameter Optimization & Larger Precision for (T)FHE", and
 * tfhe-rs core_crypto::algorithms::lwe_wopbs / lwe_private_functional_packing_keyswitch [RECALLED]):
 *
 *   extract_bits            per bit, least significant first: shift the bit under the padding bit, keyswitch to
 *                           the small key (this is the output), bootstrap with the constant accumulator
 *                           -2^(delta_log-1+i), add 2^(delta_log-1+i), subtract from the input
 *   circuit_bootstrap       per GGSW level j = 1..cbs_l: bootstrap (input + q/4) with the constant accumulator
 *                           -2^(63 - cbs_logB j), add 2^(63 - cbs_logB j)  [homomorphic_shift_boolean]; then k+1
 *                           private functional packing keyswitches, one per GGSW row (x -S_r, x 1)
 *   vertical_packing        CMUX tree over the table's polynomials with the most significant bits' GGSWs, blind
 *                           rotation by 2^i with the remaining ones (least significant first), sample extract 0
 *
 * PARITY STATUS: "parity unpinned" at ciphertext level, as for the other two oracles, and the reference
 * itself never calls this path (no caller of evaluate_encrypted_high_precision_lut in src/ or tests/; no
 * fixture, no test).  Pinned here at the decrypted level (the result is the table entry the encrypted bits
 * select, tests/test_wopbs_oracle.py) and by the two routes below agreeing bit for bit.
 *
 * Exactness.  Negacyclic products are computed two ways: schoolbook convolution in wrapping u64 arithmetic
 * (obviously correct, O(N^2)), and a Goldilocks NTT on the key split into parts of 32 or 16 bits so that every
 * exact integer sum stays below 2^63 (digits x part x N x (k+1) l); the parts are recombined mod 2^64.
 * The HIP path computes the same integers with two fp64 NTT fields + CRT.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#include "goldilocks.inc"

typedef struct {
    int32_t n, k, N;
    int32_t pbs_l, pbs_logB;   /* bootstrap (bit extraction, circuit bootstrap) */
    int32_t ks_l, ks_logB;     /* big (k*N) -> small (n) */
    int32_t pfks_l, pfks_logB; /* private functional packing keyswitch */
    int32_t cbs_l, cbs_logB;   /* levels / base of the GGSWs the circuit bootstrap produces */
    int32_t message_modulus, carry_modulus;
} orcw_params;

static int log2i(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

u64 orcw_modswitch(u64 x, int log2_2N)
{
    u64 r = (x >> (64 - log2_2N - 1)) + 1;
    return (r >> 1) & (((u64)1 << log2_2N) - 1);
}

/* signed gadget decomposition (closest representable, balanced digits), digits[0] = most significant level */
void orcw_decompose(u64 x, int logB, int l, int64_t *digits)
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

/* LWE keyswitch in_dim -> out_dim; ksk layout [in_dim][l][out_dim+1]:
 *   out = (0, ..., 0, b) - sum_t sum_j digit_j(a_t) * ksk[t][j] */
void orcw_keyswitch(int in_dim, int out_dim, int l, int logB, const u64 *ksk, const u64 *in, u64 *out)
{
    int64_t dig[64];
    memset(out, 0, sizeof(u64) * (size_t)(out_dim + 1));
    out[out_dim] = in[in_dim];
    for (int t = 0; t < in_dim; t++) {
        orcw_decompose(in[t], logB, l, dig);
        for (int j = 0; j < l; j++) {
            u64 d = (u64)dig[j];
            if (!d) continue;
            const u64 *row = ksk + ((size_t)t * l + j) * ((size_t)out_dim + 1);
            for (int c = 0; c <= out_dim; c++) out[c] -= d * row[c];
        }
    }
}

/* Private functional packing keyswitch: LWE under the big key (in_dim mask words + body) -> GLWE.  The body is an
 * input like the mask words (its key element is -1): key layout [in_dim+1][l][(k+1) N],
 *   glwe = - sum_{t <= in_dim} sum_j digit_j(in_t) * key[t][j] */
void orcw_pfpks(int in_dim, int glwe_words, int l, int logB, const u64 *key, const u64 *in, u64 *glwe)
{
    int64_t dig[64];
    memset(glwe, 0, sizeof(u64) * (size_t)glwe_words);
    for (int t = 0; t <= in_dim; t++) {
        orcw_decompose(in[t], logB, l, dig);
        for (int j = 0; j < l; j++) {
            u64 d = (u64)dig[j];
            if (!d) continue;
            const u64 *row = key + ((size_t)t * l + j) * (size_t)glwe_words;
            for (int c = 0; c < glwe_words; c++) glwe[c] -= d * row[c];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* GGSW stacks and the external product, two routes                           */
/* ------------------------------------------------------------------------- */
typedef struct {
    int N, k, l, logB, count, parts, part_bits;
    const u64 *std;   /* [count][l][k+1 rows][k+1 cols][N], borrowed */
    gl_tables *T;
    u64 *ntt;         /* [count][l][row][col][part][N], transform domain, 1/N folded in; NULL: schoolbook route */
} orcw_ggsw;

orcw_ggsw *orcw_ggsw_new(int N, int k, int l, int logB, const u64 *ggsw_std, int count, int use_ntt)
{
    orcw_ggsw *G = (orcw_ggsw *)calloc(1, sizeof(*G));
    G->N = N; G->k = k; G->l = l; G->logB = logB; G->count = count; G->std = ggsw_std;
    if (!use_ntt) return G;
    /* |sum| <= (k+1) l N (B/2) 2^part_bits must stay below 2^63 */
    int budget = 63 - (log2i((k + 1) * l) + log2i(N) + logB - 1);
    G->part_bits = budget >= 32 ? 32 : budget >= 16 ? 16 : 8;
    G->parts = 64 / G->part_bits;
    G->T = gl_tables_new(N);
    size_t polys = (size_t)count * l * (k + 1) * (k + 1);
    G->ntt = (u64 *)malloc(sizeof(u64) * polys * G->parts * N);
    u64 pmask = G->part_bits == 64 ? ~(u64)0 : (((u64)1 << G->part_bits) - 1);
    #pragma omp parallel for schedule(static)
    for (size_t q = 0; q < polys; q++)
        for (int p = 0; p < G->parts; p++) {
            u64 *dst = G->ntt + (q * G->parts + p) * N;
            for (int t = 0; t < N; t++) dst[t] = (ggsw_std[q * N + t] >> (p * G->part_bits)) & pmask;
            gl_ntt_fwd(G->T, dst);
            for (int t = 0; t < N; t++) dst[t] = gl_mul(dst[t], G->T->n_inv);
        }
    return G;
}
void orcw_ggsw_free(orcw_ggsw *G)
{
    if (!G) return;
    if (G->T) gl_tables_free(G->T);
    free(G->ntt);
    free(G);
}
int orcw_ggsw_part_bits(const orcw_ggsw *G) { return G->ntt ? G->part_bits : 0; }

/* acc += GGSW[i] (x) diff;  diff, acc: GLWE of (k+1) N words */
void orcw_extprod_add(const orcw_ggsw *G, int i, const u64 *diff, u64 *acc)
{
    int N = G->N, k1 = G->k + 1, l = G->l;
    int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * (size_t)k1 * l * N);
    int64_t tmp[64];
    for (int r = 0; r < k1; r++)
        for (int t = 0; t < N; t++) {
            orcw_decompose(diff[r * N + t], G->logB, l, tmp);
            for (int j = 0; j < l; j++) dig[((size_t)r * l + j) * N + t] = tmp[j];
        }
    if (!G->ntt) {
        const u64 *gi = G->std + (size_t)i * l * k1 * k1 * N;
        for (int j = 0; j < l; j++)
            for (int r = 0; r < k1; r++) {
                const int64_t *d = dig + ((size_t)r * l + j) * N;
                for (int c = 0; c < k1; c++) {
                    const u64 *row = gi + (((size_t)j * k1 + r) * k1 + c) * N;
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
        return;
    }
    u64 *f = (u64 *)malloc(sizeof(u64) * (size_t)k1 * l * N);
    for (size_t q = 0; q < (size_t)k1 * l * N; q++) f[q] = gl_from_i64(dig[q]);
    for (int q = 0; q < k1 * l; q++) gl_ntt_fwd(G->T, f + (size_t)q * N);
    u64 *o = (u64 *)malloc(sizeof(u64) * (size_t)N);
    const u64 *gi = G->ntt + (size_t)i * l * k1 * k1 * G->parts * N;
    for (int c = 0; c < k1; c++)
        for (int p = 0; p < G->parts; p++) {
            memset(o, 0, sizeof(u64) * (size_t)N);
            for (int j = 0; j < l; j++)
                for (int r = 0; r < k1; r++) {
                    const u64 *fr = f + ((size_t)r * l + j) * N;
                    const u64 *row = gi + (((((size_t)j * k1 + r) * k1 + c) * G->parts) + p) * N;
                    for (int t = 0; t < N; t++) o[t] = gl_add(o[t], gl_mul(fr[t], row[t]));
                }
            gl_ntt_inv(G->T, o);
            u64 *A = acc + (size_t)c * N;
            for (int t = 0; t < N; t++) {
                /* centred lift: the exact integer is below 2^63 in magnitude */
                u64 v = o[t] > GL_P / 2 ? o[t] - GL_P : o[t]; /* two's complement of the negative representative */
                A[t] += v << (p * G->part_bits);
            }
        }
    free(o);
    free(f);
    free(dig);
}

static inline u64 rot_coeff(const u64 *P, int N, int j, int a)
{
    int idx = (j - a) & (2 * N - 1);
    return idx < N ? P[idx] : (u64)0 - P[idx - N];
}

/* CMUX: c0 <- c0 + GGSW[i] (x) (c1 - c0) */
void orcw_cmux(const orcw_ggsw *G, int i, u64 *c0, const u64 *c1)
{
    size_t words = (size_t)(G->k + 1) * G->N;
    u64 *diff = (u64 *)malloc(sizeof(u64) * words);
    for (size_t q = 0; q < words; q++) diff[q] = c1[q] - c0[q];
    orcw_extprod_add(G, i, diff, c0);
    free(diff);
}

/* programmable bootstrap: small LWE (n+1) -> big LWE (k*N+1), test polynomial tv (N words); bsk: n GGSWs */
void orcw_bootstrap(const orcw_ggsw *bsk, int n, const u64 *lwe, const u64 *tv, u64 *out_big)
{
    int N = bsk->N, k = bsk->k, k1 = k + 1;
    int log2_2N = log2i(2 * N);
    u64 *acc = (u64 *)calloc((size_t)k1 * N, sizeof(u64));
    u64 *diff = (u64 *)malloc(sizeof(u64) * (size_t)k1 * N);
    int bt = (int)orcw_modswitch(lwe[n], log2_2N);
    for (int j = 0; j < N; j++) acc[(size_t)k * N + j] = rot_coeff(tv, N, j, (2 * N - bt) & (2 * N - 1));
    for (int i = 0; i < n; i++) {
        int a = (int)orcw_modswitch(lwe[i], log2_2N);
        if (a == 0) continue;
        for (int r = 0; r < k1; r++)
            for (int j = 0; j < N; j++)
                diff[r * N + j] = rot_coeff(acc + (size_t)r * N, N, j, a) - acc[(size_t)r * N + j];
        orcw_extprod_add(bsk, i, diff, acc);
    }
    for (int r = 0; r < k; r++) {
        const u64 *A = acc + (size_t)r * N;
        out_big[r * N] = A[0];
        for (int t = 1; t < N; t++) out_big[r * N + t] = (u64)0 - A[N - t];
    }
    out_big[k * N] = acc[(size_t)k * N];
    free(acc);
    free(diff);
}

/* bootstrap with a constant accumulator: every coefficient = value (negacyclic: the sign of the phase) */
static void bootstrap_const(const orcw_ggsw *bsk, int n, const u64 *lwe, u64 value, u64 *out_big)
{
    u64 *tv = (u64 *)malloc(sizeof(u64) * (size_t)bsk->N);
    for (int j = 0; j < bsk->N; j++) tv[j] = value;
    orcw_bootstrap(bsk, n, lwe, tv, out_big);
    free(tv);
}

/* ------------------------------------------------------------------------- */
/* The three stages                                                           */
/* ------------------------------------------------------------------------- */

/* extract_bits: `nb` bits from position delta_log upwards of the plaintext of in_big (k*N+1 words).
 * out_small: nb rows of n+1 words, row 0 = MOST significant extracted bit, each encrypting bit * 2^63. */
void orcw_extract_bits(const orcw_params *P, const orcw_ggsw *bsk, const u64 *ksk, int delta_log, int nb,
                       const u64 *in_big, u64 *out_small)
{
    int kN = P->k * P->N, n = P->n;
    size_t brow = (size_t)kN + 1, srow = (size_t)n + 1;
    u64 *buf = (u64 *)malloc(sizeof(u64) * brow), *shifted = (u64 *)malloc(sizeof(u64) * brow);
    u64 *pbs = (u64 *)malloc(sizeof(u64) * brow), *ks = (u64 *)malloc(sizeof(u64) * srow);
    memcpy(buf, in_big, sizeof(u64) * brow);
    for (int i = 0; i < nb; i++) {
        const u64 shift = (u64)1 << (64 - delta_log - i - 1); /* bit i of the message under the padding bit */
        for (size_t q = 0; q < brow; q++) shifted[q] = buf[q] * shift;
        orcw_keyswitch(kN, n, P->ks_l, P->ks_logB, ksk, shifted, ks);
        memcpy(out_small + srow * (size_t)(nb - 1 - i), ks, sizeof(u64) * srow);
        if (i == nb - 1) break;
        ks[n] += (u64)1 << 62; /* q/4: centre the error for the negacyclic sign function */
        const u64 alpha = (u64)1 << (delta_log - 1 + i);
        bootstrap_const(bsk, n, ks, (u64)0 - alpha, pbs);
        pbs[kN] += alpha; /* 0 if the bit was 0, 2 alpha = bit weight otherwise */
        for (size_t q = 0; q < brow; q++) buf[q] -= pbs[q];
    }
    free(buf); free(shifted); free(pbs); free(ks);
}

/* circuit_bootstrap_boolean: small LWE encrypting bit * 2^63 -> GGSW of the bit, standard domain,
 * layout [cbs_l][k+1 rows][(k+1) N] (level 1 first = the layout of a bootstrapping-key entry).
 * pfpksk: [k+1][k*N+1][pfks_l][(k+1) N]. */
void orcw_circuit_bootstrap(const orcw_params *P, const orcw_ggsw *bsk, const u64 *pfpksk, const u64 *in_small,
                            u64 *ggsw_out)
{
    int kN = P->k * P->N, n = P->n, k1 = P->k + 1;
    size_t brow = (size_t)kN + 1, srow = (size_t)n + 1, glwe = (size_t)k1 * P->N;
    size_t key_words = brow * P->pfks_l * glwe;
    u64 *in = (u64 *)malloc(sizeof(u64) * srow), *pbs = (u64 *)malloc(sizeof(u64) * brow);
    memcpy(in, in_small, sizeof(u64) * srow);
    in[n] += (u64)1 << 62;
    for (int j = 0; j < P->cbs_l; j++) {
        const u64 alpha = (u64)1 << (63 - P->cbs_logB * (j + 1));
        bootstrap_const(bsk, n, in, (u64)0 - alpha, pbs);
        pbs[kN] += alpha; /* bit * 2^(64 - cbs_logB (j+1)) */
        for (int r = 0; r < k1; r++)
            orcw_pfpks(kN, (int)glwe, P->pfks_l, P->pfks_logB, pfpksk + (size_t)r * key_words, pbs,
                       ggsw_out + ((size_t)j * k1 + r) * glwe);
    }
    free(in); free(pbs);
}

/* vertical_packing: `bits` GGSWs (index 0 = most significant bit) select entry v of `lut`
 * (max(2^bits, N) words, entry v at index v); out_big = LWE of lut[v] under the GLWE key. */
void orcw_vertical_packing(const orcw_ggsw *G, int bits, const u64 *lut, u64 *out_big)
{
    int N = G->N, k = G->k, k1 = k + 1, logN = log2i(N);
    int tree = bits > logN ? bits - logN : 0;
    size_t glwe = (size_t)k1 * N;
    size_t polys = (size_t)1 << tree;
    u64 *cur = (u64 *)calloc(polys * glwe, sizeof(u64)); /* trivial GLWEs: zero mask, body = table polynomial */
    for (size_t q = 0; q < polys; q++) memcpy(cur + q * glwe + (size_t)k * N, lut + q * N, sizeof(u64) * (size_t)N);
    /* CMUX tree: the last of the tree's GGSWs (lowest tree bit) pairs neighbouring polynomials */
    for (int lev = 0; lev < tree; lev++) {
        int g = tree - 1 - lev;
        polys >>= 1;
        for (size_t q = 0; q < polys; q++) {
            orcw_cmux(G, g, cur + 2 * q * glwe, cur + (2 * q + 1) * glwe);
            if (q) memcpy(cur + q * glwe, cur + 2 * q * glwe, sizeof(u64) * glwe);
        }
    }
    /* blind rotation, least significant bit first: c <- CMUX(bit, c, c * X^(-2^i)) */
    u64 *rot = (u64 *)malloc(sizeof(u64) * glwe);
    int deg = 1;
    for (int g = bits - 1; g >= tree; g--) {
        for (int r = 0; r < k1; r++)
            for (int j = 0; j < N; j++) rot[r * N + j] = rot_coeff(cur + (size_t)r * N, N, j, (2 * N - deg) & (2 * N - 1));
        orcw_cmux(G, g, cur, rot);
        deg <<= 1;
    }
    for (int r = 0; r < k; r++) {
        const u64 *A = cur + (size_t)r * N;
        out_big[r * N] = A[0];
        for (int t = 1; t < N; t++) out_big[r * N + t] = (u64)0 - A[N - t];
    }
    out_big[k * N] = cur[(size_t)k * N];
    free(rot); free(cur);
}

u64 orcw_phase(int dim, const u64 *sk_bits, const u64 *ct)
{
    u64 ph = ct[dim];
    for (int i = 0; i < dim; i++) if (sk_bits[i]) ph -= ct[i];
    return ph;
}
