/*
 * helm_wopbs.h — C ABI of the MI355X-native WoP-PBS wide-LUT path (bit extraction, circuit
 * bootstrap, vertical packing) on top of the LUT-mode engine of helm_shortint.h.
 *
 * Reference interfaces this replaces (arithmetic inside the `tfhe` 0.4.1 crate, Cargo.toml:18,
 * absent from the tree; the reference itself never calls this path - SURVEY.md A3/N5):
 *   src/gates.rs:721-742   Gate::evaluate_encrypted_high_precision_lut
 *   src/gates.rs:787-815   high_precision_lut(): blocks -> radix ciphertext (first input = most
 *                          significant block), WopbsKey::keyswitch_to_wopbs_params, the table,
 *                          WopbsKey::wopbs, keyswitch_to_pbs_params, block 0 returned
 *   src/gates.rs:817-864   generate_high_precision_lut_radix_helm: the table of a wide LUT
 *
 * Shape of the replacement.  A WoP-PBS context sits beside a LUT-mode context (the "PBS side":
 * helm_si_ctx, its keys, its wire table); it owns a second parameter set (the "WoP side") and five
 * keys.  helm_wop_eval_luts() evaluates a batch of wide LUT gates, every stage batched over the
 * whole batch:
 *   1  cleaning bootstrap of every input block (PBS side, identity table)     keyswitch_to_wopbs_params
 *   2  keyswitch PBS-side big key -> WoP-side big key                          (ksk_pbs_to_wopbs)
 *   3  bit extraction: bits_per_block bits per block, each: shift, keyswitch to the WoP small key,
 *      and for all but the last bit a bootstrap that removes the bit          WopbsKey::wopbs
 *   4  circuit bootstrap: per bit cbs_l bootstraps, then (k+1) cbs_l private functional packing
 *      keyswitches -> one GGSW per bit, converted to the transform domain
 *   5  vertical packing: CMUX tree over the table's polynomials (bits above log2 N), blind rotation
 *      by 2^i with the low bits' GGSWs, sample extract
 *   6  keyswitch WoP-side big key -> PBS-side small key, bootstrap (identity)  keyswitch_to_pbs_params
 * The reference computes one table per radix block and returns block 0 (gates.rs:814); only block 0's
 * table is evaluated here.
 *
 * Conventions as in helm_hip.h: 0 / negative helm_status, helm_hip_last_error(), caller owns host
 * buffers, no CPU fallback.  All words uint64_t, arithmetic mod 2^64.
 *
 * Layouts
 *   bootstrapping key      [n][pbs_l][k+1][k+1][N]                  GGSW(s_i) under the WoP GLWE key
 *   keyswitching key       [k*N][ks_l][n+1]                          WoP big -> WoP small
 *   ksk_pbs_to_wopbs       [k_p*N_p][l][k*N+1]                       PBS-side big -> WoP big
 *   ksk_wopbs_to_pbs       [k*N][l][n_p+1]                           WoP big -> PBS-side small
 *   pfpksk                 [k+1][k*N+1][pfks_l][(k+1) N]             key r: x (-S_r) for r < k, x 1 for r = k;
 *                          input element k*N is the body (key element -1); rows are GLWE ciphertexts
 *                          (mask polynomials, body polynomial)
 *   GGSW of a bit          [cbs_l][k+1][k+1][N] standard domain (the layout of a bootstrapping-key entry)
 *   table of a gate        max(2^total_bits, N) words, entry v at index v, already scaled (value * delta)
 */
#ifndef HELM_WOPBS_H
#define HELM_WOPBS_H

#include <stddef.h>
#include <stdint.h>
#include "helm_shortint.h"
#include "helm_client.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tfhe::shortint::WopbsParameters, runtime values */
typedef struct {
    int32_t n, k, N;
    int32_t pbs_l, pbs_logB;
    int32_t ks_l, ks_logB;
    int32_t pfks_l, pfks_logB;
    int32_t cbs_l, cbs_logB;
    int32_t message_modulus, carry_modulus;
} helm_wop_params;

typedef struct helm_wop_ctx helm_wop_ctx;

/* WopbsKey::new_wopbs_key: the context borrows `pbs_side` (which must outlive it and have its keys loaded
 * before helm_wop_eval_luts) and creates the WoP-side engine on the same device and stream. */
int helm_wop_ctx_create(helm_si_ctx *pbs_side, const helm_wop_params *params, helm_wop_ctx **out);
int helm_wop_ctx_destroy(helm_wop_ctx *ctx);
int helm_wop_get_params(const helm_wop_ctx *ctx, helm_wop_params *out);

enum {
    HELM_WOP_KEY_BSK = 0,          /* WoP-side bootstrapping key */
    HELM_WOP_KEY_KSK = 1,          /* WoP big -> WoP small */
    HELM_WOP_KEY_KSK_TO_WOPBS = 2, /* PBS-side big -> WoP big; decomposition given to the call */
    HELM_WOP_KEY_KSK_TO_PBS = 3,   /* WoP big -> PBS-side small; decomposition given to the call */
    HELM_WOP_KEY_PFPKSK = 4,       /* the k+1 private functional packing keyswitching keys */
    HELM_WOP_KEY_LWE_SECRET = 5,   /* client only: n words of 0/1 */
    HELM_WOP_KEY_GLWE_SECRET = 6   /* client only: k*N words of 0/1 */
};
/* l, logB: decomposition of keys 2 and 3 (ignored for the others, whose decomposition is in the parameters) */
int helm_wop_load_key(helm_wop_ctx *ctx, int which, const uint64_t *words, size_t n_words, int32_t l, int32_t logB);

/* generate_high_precision_lut_radix_helm (gates.rs:817-864) for block 0: n_blocks blocks of bits_per_block bits,
 * basis = message_modulus.  Index v of the table: field j (bits [j*b, (j+1)*b) of v) is the value of block j;
 * x = sum_j field_j * basis^j  mod basis^n_blocks;  entry = ((truth[x] & 1) mod basis) * delta.  truth: the gate's
 * table, one word per entry (gates.rs:746-748; an index past truth_len reads 0 - the reference would panic).
 * table_out: helm_wop_table_words(params, n_blocks * bits_per_block) words. */
size_t helm_wop_table_words(const helm_wop_params *params, int32_t total_bits);
int helm_wop_make_table(const helm_wop_params *params, int32_t n_blocks, int32_t bits_per_block, const uint64_t *truth,
                        size_t truth_len, uint64_t *table_out);

/* A batch of `count` wide LUT gates with n_inputs inputs each: high_precision_lut() per gate.
 * in_idx [count][n_inputs] rows of `w` (first input = most significant block, gates.rs:795-799);
 * bits_per_block: bits extracted per block - log2(message_modulus * carry_modulus) is what tfhe's degree
 * bookkeeping extracts after the cleaning bootstrap; 1 is enough when every input holds a single bit;
 * tables [count][helm_wop_table_words(n_inputs * bits_per_block)];  out_idx [count] rows of `w`.
 * n_inputs * bits_per_block <= log2(N) + 6.
 * The gates of a call are independent (one netlist level): an output row that is an input row of ANOTHER gate of the
 * call is refused with HELM_ERR_INVALID - the batch runs in chunks, so such a call would depend on the chunking.  A gate
 * may write over one of its own inputs.
 * Multi-GPU: after helm_si_set_exchange() on the PBS-side context a batch of at least min_batch gates is split over the
 * ranks by gate (every stage of a gate on one rank), the result rows go through the same all-gather callback and are
 * scattered into every rank's table; identical ciphertexts to the unsharded call. */
int helm_wop_eval_luts(helm_wop_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, int32_t n_inputs,
                       int32_t bits_per_block, const uint64_t *tables, const int32_t *out_idx, int64_t count);

/* Stage primitives on host buffers (tests).
 * extract_bits: in_big count x (k*N+1) under the WoP big key -> out_small count x nb x (n+1), bit 0 of a row =
 *   least significant extracted bit (position delta_log), each encrypting bit * 2^63.
 * circuit_bootstrap: in_small count x (n+1), bit * 2^63 -> ggsw_out count x [cbs_l][k+1][k+1][N].
 * vertical_packing: ggsw count x bits x [cbs_l][k+1][k+1][N] (index 0 = least significant bit),
 *   tables count x helm_wop_table_words(bits) -> out_big count x (k*N+1) under the WoP big key. */
int helm_wop_extract_bits_batch(helm_wop_ctx *ctx, const uint64_t *in_big, int32_t delta_log, int32_t nb,
                                uint64_t *out_small, int64_t count);
int helm_wop_circuit_bootstrap_batch(helm_wop_ctx *ctx, const uint64_t *in_small, uint64_t *ggsw_out, int64_t count);
int helm_wop_vertical_packing_batch(helm_wop_ctx *ctx, const uint64_t *ggsw, int32_t bits, const uint64_t *tables,
                                    uint64_t *out_big, int64_t count);

typedef struct {
    double clean_ms, to_wopbs_ms, extract_ms, cbs_pbs_ms, pfpks_ms, convert_ms, packing_ms, to_pbs_ms;
    int64_t gates, bootstraps;
} helm_wop_timing;
/* accumulated over helm_wop_eval_luts calls since the last reset (HIP events; the call synchronises) */
int helm_wop_get_timing(helm_wop_ctx *ctx, helm_wop_timing *out, int reset);

/* ---- client side (CPU): WopbsKey::new_wopbs_key's key material ------------------------------------
 * "wopbs_m1c1": WOPBS_PARAM_MESSAGE_1_CARRY_1_KS_PBS, "wopbs_m2c2": WOPBS_PARAM_MESSAGE_2_CARRY_2_KS_PBS
 * [dimensions recalled]; "wop_toy_*": small sets for exact oracle comparisons. */
typedef struct helm_wop_client_key helm_wop_client_key;
int helm_wop_client_named_params(const char *name, helm_wop_params *params, double *lwe_noise_std,
                                 double *glwe_noise_std);
/* keys 2 and 3 use the PBS side's keyswitch decomposition (ks_l, ks_logB of pbs_key's parameters) */
int helm_wop_client_keygen(const helm_si_client_key *pbs_key, const helm_wop_params *params, double lwe_noise_std,
                           double glwe_noise_std, uint64_t seed, helm_wop_client_key **out);
void helm_wop_client_key_free(helm_wop_client_key *key);
int helm_wop_client_key_part(const helm_wop_client_key *key, int which, const uint64_t **words, size_t *n_words);

#ifdef __cplusplus
}
#endif
#endif /* HELM_WOPBS_H */
