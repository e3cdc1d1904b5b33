/*
 * shim_sequence_si.c — the call sequences rust/helm-hip's HipLutCircuit (src/lut.rs) and HipArithCircuit
 * (src/arith.rs) make, from C.
 *
 * The Rust shim cannot be compiled in this image (no rustc).  This program issues the same C-ABI calls in the same
 * order with the same argument shapes, against libhelm_hip.so and libhelm_host.so only:
 *
 *   LUT mode (reference src/circuit.rs:969-1111, gates::lut() src/gates.rs:754-785) on the 8-bit adder of 3-input LUTs
 *   (tests/circuit_test.rs:266-311 uses its 2-input sibling): new -> keys through the tfhe-order converters ->
 *   encrypt_inputs (alloc, create_trivial(0), upload) -> evaluate_encrypted (one helm_si_eval_lut_level per netlist level)
 *   -> evaluate_ready shape (one more level of 3-input look-ups) -> decrypt_outputs -> Drop.
 *
 *   Arithmetic mode (reference src/circuit.rs:1113-1483, FheUintN operators src/gates.rs:306-702) on the FheUint16 known
 *   answers of tests/gates_test.rs:127-310 (K-7): 10+20, 20-10, 10*20, then 30+40, 40-30, 30*40: encrypt radix blocks ->
 *   helm_host_radix_scratch_rows -> one helm_host_radix_level per level -> download -> decrypt.
 *
 * Prints "ok" and returns 0 when every decrypted value matches.  Usage: shim_sequence_si [shortint parameter set]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "helm_client.h"
#include "helm_host.h"
#include "helm_shortint.h"

#define CHECK(call, what)                                                                              \
    do {                                                                                               \
        int rc__ = (call);                                                                             \
        if (rc__ != 0) {                                                                               \
            fprintf(stderr, "%s failed (%d): hip='%s' host='%s' keys='%s'\n", what, rc__,              \
                    helm_hip_last_error(), helm_host_last_error(), helm_keys_last_error());            \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

/* rows of the LUT adder: a[0..7] = 0..7, b[0..7] = 8..15, cin = 16, sum[0..7] = 17..24, c1..c7 = 25..31, cout = 32 */
enum { ROW_A = 0, ROW_B = 8, ROW_CIN = 16, ROW_SUM = 17, ROW_C = 24 /* c_i at ROW_C + i, i = 1..8 (c8 = cout) */, LUT_ROWS = 33 };

static int lut_mode(helm_si_client_key *ck, helm_si_ctx *ctx, const helm_si_params *P)
{
    const size_t row = (size_t)P->k * P->N + 1;
    const unsigned a = 0xB7, b = 0x6E, cin = 1;
    /* encrypt_inputs (circuit.rs:970-1000): every gate output <- create_trivial(0), inputs <- client_key.encrypt(bit) */
    helm_si_wires *w = NULL;
    CHECK(helm_si_wires_alloc(ctx, LUT_ROWS, &w), "helm_si_wires_alloc");
    int32_t triv[16];
    uint64_t zero[16] = {0};
    for (int i = 0; i < 16; i++) triv[i] = ROW_SUM + i;
    CHECK(helm_si_wires_set_trivial(ctx, w, triv, zero, 16), "helm_si_wires_set_trivial");
    int32_t in_rows[17];
    uint64_t in_vals[17];
    for (int i = 0; i < 8; i++) {
        in_rows[i] = ROW_A + i;     in_vals[i] = (a >> i) & 1;
        in_rows[8 + i] = ROW_B + i; in_vals[8 + i] = (b >> i) & 1;
    }
    in_rows[16] = ROW_CIN; in_vals[16] = cin;
    uint64_t *cts = malloc(17 * row * 8);
    CHECK(helm_si_client_encrypt(ck, in_vals, 17, cts), "helm_si_client_encrypt");
    CHECK(helm_si_wires_upload(ctx, w, in_rows, cts, 17), "helm_si_wires_upload");
    /* evaluate_encrypted (circuit.rs:1032-1083): one call per level; level i holds sum[i] (0x96) and c_{i+1} (0xE8) */
    for (int i = 0; i < 8; i++) {
        const int32_t carry_in = i == 0 ? ROW_CIN : ROW_C + i;
        const int32_t arity[2] = {3, 3};
        const int32_t in_idx[6] = {ROW_A + i, ROW_B + i, carry_in, ROW_A + i, ROW_B + i, carry_in};
        const uint64_t table[2] = {0x96, 0xE8};
        const int32_t out_idx[2] = {ROW_SUM + i, ROW_C + i + 1};
        CHECK(helm_si_eval_lut_level(ctx, w, arity, in_idx, 3, table, out_idx, 2), "helm_si_eval_lut_level");
    }
    /* evaluate_ready shape (circuit.rs:1002-1030; here one 3-input look-up per output):
     * valid = READY ? new : valid with READY = cout, new = sum[0], valid = sum[1] -> written over sum[1]'s spare c row */
    {
        const int32_t arity[1] = {3}, in_idx[3] = {ROW_C + 8, ROW_SUM + 0, ROW_SUM + 1}, out_idx[1] = {ROW_C + 1};
        const uint64_t table[1] = {0xCA}; /* index = READY<<2 | new<<1 | old: READY ? new : old */
        CHECK(helm_si_eval_lut_level(ctx, w, arity, in_idx, 3, table, out_idx, 1), "helm_si_eval_lut_level (ready)");
    }
    CHECK(helm_si_sync(ctx), "helm_si_sync");
    /* decrypt_outputs (circuit.rs:1085-1110) */
    int32_t out_rows[10];
    for (int i = 0; i < 8; i++) out_rows[i] = ROW_SUM + i;
    out_rows[8] = ROW_C + 8;
    out_rows[9] = ROW_C + 1;
    uint64_t *dl = malloc(10 * row * 8), vals[10];
    CHECK(helm_si_wires_download(ctx, w, out_rows, dl, 10), "helm_si_wires_download");
    CHECK(helm_si_client_decrypt(ck, dl, 10, vals), "helm_si_client_decrypt");
    unsigned total = 0;
    for (int i = 0; i < 9; i++) total |= (unsigned)(vals[i] % (uint64_t)P->message_modulus) << i;
    const unsigned cout = (a + b + cin) >> 8 & 1, s0 = (a + b + cin) & 1, s1 = (a + b + cin) >> 1 & 1;
    const unsigned latched = cout ? s0 : s1;
    CHECK(helm_si_wires_free(ctx, w), "helm_si_wires_free");
    free(cts); free(dl);
    if (total != a + b + cin || vals[9] % (uint64_t)P->message_modulus != latched) {
        fprintf(stderr, "LUT adder: got %u (latch %llu), expected %u (latch %u)\n", total, (unsigned long long)vals[9], a + b + cin, latched);
        return 1;
    }
    return 0;
}

/* FheUint16 = 8 blocks of 2 message bits; integers at rows A = 0, B = 8, S = 16, D = 24, P = 32; scratch from 40 */
enum { BLOCKS = 8, INT_A = 0, INT_B = 8, INT_S = 16, INT_D = 24, INT_P = 32, INT_ROWS = 40 };

static int arith_mode(helm_si_client_key *ck, helm_si_ctx *ctx, const helm_si_params *P)
{
    const size_t row = (size_t)P->k * P->N + 1;
    /* one level: add g0(A, B, S); sub g1(B, A, D); mult g2(A, B, P)  (gates_test.rs:127-190) */
    const helm_radix_op ops[3] = {{HELM_RADIX_ADD, INT_A, INT_B, INT_S, 0, 0},
                                  {HELM_RADIX_SUB, INT_B, INT_A, INT_D, 0, 0},
                                  {HELM_RADIX_MUL, INT_A, INT_B, INT_P, 0, 0}};
    const int64_t scratch = helm_host_radix_scratch_rows(ctx, BLOCKS, ops, 3);
    if (scratch < 0) { fprintf(stderr, "radix_scratch_rows: %s\n", helm_host_last_error()); return 1; }
    helm_si_wires *w = NULL;
    CHECK(helm_si_wires_alloc(ctx, INT_ROWS + scratch, &w), "helm_si_wires_alloc");
    uint64_t *cts = malloc(2 * BLOCKS * row * 8), *dl = malloc(3 * BLOCKS * row * 8);
    const unsigned kat[2][2] = {{10, 20}, {30, 40}};
    for (int t = 0; t < 2; t++) {
        const unsigned x = kat[t][0], y = kat[t][1];
        /* FheUint16::try_encrypt: block i holds bits 2i+1..2i, least significant block first */
        int32_t rows[2 * BLOCKS];
        uint64_t digits[2 * BLOCKS];
        for (int i = 0; i < BLOCKS; i++) {
            rows[i] = INT_A + i;          digits[i] = (x >> (2 * i)) & 3;
            rows[BLOCKS + i] = INT_B + i; digits[BLOCKS + i] = (y >> (2 * i)) & 3;
        }
        CHECK(helm_si_client_encrypt(ck, digits, 2 * BLOCKS, cts), "helm_si_client_encrypt");
        CHECK(helm_si_wires_upload(ctx, w, rows, cts, 2 * BLOCKS), "helm_si_wires_upload");
        int64_t pbs = 0, rounds = 0;
        CHECK(helm_host_radix_level(ctx, w, BLOCKS, ops, 3, INT_ROWS, &pbs, &rounds), "helm_host_radix_level");
        CHECK(helm_si_sync(ctx), "helm_si_sync");
        int32_t out_rows[3 * BLOCKS];
        uint64_t vals[3 * BLOCKS];
        for (int i = 0; i < 3 * BLOCKS; i++) out_rows[i] = INT_S + i;
        CHECK(helm_si_wires_download(ctx, w, out_rows, dl, 3 * BLOCKS), "helm_si_wires_download");
        CHECK(helm_si_client_decrypt(ck, dl, 3 * BLOCKS, vals), "helm_si_client_decrypt");
        unsigned got[3] = {0, 0, 0};
        for (int q = 0; q < 3; q++)
            for (int i = 0; i < BLOCKS; i++) got[q] |= (unsigned)(vals[q * BLOCKS + i] % 4) << (2 * i);
        const unsigned want[3] = {(x + y) & 0xFFFF, (y - x) & 0xFFFF, (x * y) & 0xFFFF};
        if (memcmp(got, want, sizeof got) != 0 || pbs <= 0 || rounds <= 0) {
            fprintf(stderr, "FheUint16 (%u, %u): got %u %u %u, expected %u %u %u (%lld bootstraps)\n", x, y, got[0], got[1], got[2],
                    want[0], want[1], want[2], (long long)pbs);
            return 1;
        }
    }
    CHECK(helm_si_wires_free(ctx, w), "helm_si_wires_free");
    free(cts); free(dl);
    return 0;
}

int main(int argc, char **argv)
{
    const char *set = argc > 1 ? argv[1] : "shortint_m2c2";
    helm_si_params P;
    double lwe_std, glwe_std;
    CHECK(helm_si_client_named_params(set, &P, &lwe_std, &glwe_std), "helm_si_client_named_params");
    /* client side: shortint::gen_keys(PARAM_...) (helm.rs:301) / generate_keys(config) (helm.rs:88); OS entropy */
    helm_si_client_key *ck = NULL;
    CHECK(helm_si_client_keygen(&P, lwe_std, glwe_std, HELM_SEED_OS_ENTROPY, &ck), "helm_si_client_keygen");
    /* keys::standard_keys64(): the words arrive in tfhe's container order and go through the converters */
    const size_t nb = helm_si_client_bsk_words(ck), nk = helm_si_client_ksk_words(ck);
    uint64_t *t_bsk = malloc(nb * 8), *t_ksk = malloc(nk * 8), *bsk = malloc(nb * 8), *ksk = malloc(nk * 8);
    if (P.grouping_factor > 1) {
        /* multi-bit sets (helm.rs:83): tfhe's own multi-bit key layout is not importable word for word (the converter
         * refuses it); the shim generates the bootstrapping key in this ABI's subset-indicator convention (keys.rs) */
        memcpy(bsk, helm_si_client_bsk(ck), nb * 8);
    } else {
        CHECK(helm_keys_bsk64_to_tfhe(&P, helm_si_client_bsk(ck), t_bsk, nb), "to tfhe order (bsk)");
        CHECK(helm_keys_bsk64_from_tfhe(&P, t_bsk, bsk, nb), "helm_keys_bsk64_from_tfhe");
    }
    CHECK(helm_keys_ksk64_to_tfhe(&P, helm_si_client_ksk(ck), t_ksk, nk), "to tfhe order (ksk)");
    CHECK(helm_keys_ksk64_from_tfhe(&P, t_ksk, ksk, nk), "helm_keys_ksk64_from_tfhe");
    /* HipLutCircuit::new / HipArithCircuit::new */
    helm_si_ctx *ctx = NULL;
    CHECK(helm_si_ctx_create(0, &P, &ctx), "helm_si_ctx_create");
    CHECK(helm_si_load_bootstrap_key(ctx, bsk, nb), "helm_si_load_bootstrap_key");
    CHECK(helm_si_load_keyswitch_key(ctx, ksk, nk), "helm_si_load_keyswitch_key");
    free(t_bsk); free(t_ksk); free(bsk); free(ksk);
    if (lut_mode(ck, ctx, &P)) return 1;
    if (arith_mode(ck, ctx, &P)) return 1;
    /* Drop */
    CHECK(helm_si_ctx_destroy(ctx), "helm_si_ctx_destroy");
    helm_si_client_key_free(ck);
    printf("ok: %s, 8-bit LUT adder 0xB7 + 0x6E + 1 = 0x126 and its READY latch; FheUint16 10+20, 20-10, 10*20, 30+40, 40-30, 30*40\n", set);
    return 0;
}
