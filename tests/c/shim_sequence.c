/*
 * shim_sequence.c — the exact call sequence rust/helm-hip makes, from C.
 *
 * The Rust shim cannot be compiled in this image (no rustc).  This program is the same sequence of C-ABI
 * calls in the same order with the same argument shapes - HipGateCircuit::new, encrypt_inputs (alloc,
 * set_trivial, upload), build_program (pack_levels + program_create), evaluate_encrypted (program_run + sync),
 * evaluate_ready (one MUX level), decrypt_outputs (download), Drop - on the 2-bit adder of reference
 * tests/circuit_test.rs:17-45 (all inputs true => sum[0] = sum[1] = cout = 1), with keys taken through the
 * import path the shim uses (tfhe container order -> helm_keys_*_from_tfhe).  Links libhelm_hip.so and
 * libhelm_host.so only; prints "ok" and returns 0 when every decrypted output matches.
 * Usage: shim_sequence [parameter set name]   (default boolean_default)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "helm_client.h"
#include "helm_hip.h"
#include "helm_host.h"

#define CHECK(call, what)                                                                       \
    do {                                                                                        \
        int rc__ = (call);                                                                      \
        if (rc__ != 0) {                                                                        \
            fprintf(stderr, "%s failed (%d): hip='%s' client='%s' keys='%s'\n", what, rc__,     \
                    helm_hip_last_error(), helm_client_last_error(), helm_keys_last_error());   \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

/* 2-bit ripple adder, 10 gates (reference tests/verilog_parser_test.rs:9-11): rows 0..4 = a0 a1 b0 b1 cin,
 * 5.. = gate outputs.  Levels as Circuit::compute_levels gives them. */
enum { A0, A1, B0, B1, CIN, I0, S0, T0, T1, C1, I1, S1, T2, T3, COUT, N_WIRES };
static const int32_t OP[10] = {HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_XOR, HELM_GATE_AND, /* level 1 */
                               HELM_GATE_XOR, HELM_GATE_AND,                               /* level 2 */
                               HELM_GATE_OR,                                               /* level 3 */
                               HELM_GATE_XOR, HELM_GATE_AND,                               /* level 4 */
                               HELM_GATE_OR};                                              /* level 5 */
static const int32_t IN0[10] = {A0, A0, A1, A1, I0, I0, T0, I1, I1, T2};
static const int32_t IN1[10] = {B0, B0, B1, B1, CIN, CIN, T1, C1, C1, T3};
static const int32_t IN2[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static const int32_t OUT[10] = {I0, T0, I1, T2, S0, T1, C1, S1, T3, COUT};

int main(int argc, char **argv)
{
    const char *set = argc > 1 ? argv[1] : "boolean_default";
    helm_hip_params P;
    double lwe_std, glwe_std;
    CHECK(helm_client_named_params(set, &P, &lwe_std, &glwe_std), "named_params");

    /* client side: HELM's gen_keys() (helm.rs:241); OS entropy */
    helm_client_key *ck = NULL;
    CHECK(helm_client_keygen(&P, lwe_std, glwe_std, HELM_SEED_OS_ENTROPY, &ck), "keygen");

    /* keys::standard_keys(): the words arrive in tfhe's container order and go through the converters */
    const size_t nb = helm_client_bsk_words(ck), nk = helm_client_ksk_words(ck);
    uint32_t *t_bsk = malloc(nb * 4), *t_ksk = malloc(nk * 4), *bsk = malloc(nb * 4), *ksk = malloc(nk * 4);
    CHECK(helm_keys_bsk32_to_tfhe(&P, helm_client_bsk(ck), t_bsk, nb), "to tfhe order (bsk)");
    CHECK(helm_keys_ksk32_to_tfhe(&P, helm_client_ksk(ck), t_ksk, nk), "to tfhe order (ksk)");
    CHECK(helm_keys_bsk32_from_tfhe(&P, t_bsk, bsk, nb), "helm_keys_bsk32_from_tfhe");
    CHECK(helm_keys_ksk32_from_tfhe(&P, t_ksk, ksk, nk), "helm_keys_ksk32_from_tfhe");

    /* HipGateCircuit::new */
    helm_hip_ctx *ctx = NULL;
    CHECK(helm_hip_ctx_create(0, &P, &ctx), "helm_hip_ctx_create");
    CHECK(helm_hip_load_bootstrap_key(ctx, bsk, nb), "helm_hip_load_bootstrap_key");
    CHECK(helm_hip_load_keyswitch_key(ctx, ksk, nk), "helm_hip_load_keyswitch_key");

    /* encrypt_inputs: gate outputs <- trivial(false), inputs <- encrypt(true) */
    helm_hip_wires *wires = NULL;
    CHECK(helm_hip_wires_alloc(ctx, N_WIRES, &wires), "helm_hip_wires_alloc");
    int32_t triv[10];
    uint8_t zeros[10] = {0};
    for (int g = 0; g < 10; g++) triv[g] = OUT[g];
    CHECK(helm_hip_wires_set_trivial(ctx, wires, triv, zeros, 10), "helm_hip_wires_set_trivial");
    const size_t row = (size_t)P.n + 1;
    const int32_t in_rows[5] = {A0, A1, B0, B1, CIN};
    const uint8_t in_bits[5] = {1, 1, 1, 1, 1};
    uint32_t *cts = malloc(5 * row * 4);
    CHECK(helm_client_encrypt_bool(ck, in_bits, 5, cts), "encrypt");
    CHECK(helm_hip_wires_upload(ctx, wires, in_rows, cts, 5), "helm_hip_wires_upload");

    /* build_program: level map -> packed launches -> program */
    const int64_t off[6] = {0, 4, 6, 7, 9, 10}; /* levels of the adder */
    int32_t op[10], i0[10], i1[10], i2[10], out[10];
    int64_t order[10], poff[11], n_launch = 0;
    const int64_t q = helm_hip_launch_quantum(ctx);
    if (q <= 0) { fprintf(stderr, "launch_quantum: %s\n", helm_hip_last_error()); return 1; }
    if (helm_host_pack_levels(OP, IN0, IN1, IN2, OUT, off, 5, q, order, poff, &n_launch) < 0) {
        fprintf(stderr, "pack_levels: %s\n", helm_host_last_error());
        return 1;
    }
    for (int g = 0; g < 10; g++) {
        op[g] = OP[order[g]]; i0[g] = IN0[order[g]]; i1[g] = IN1[order[g]]; i2[g] = IN2[order[g]]; out[g] = OUT[order[g]];
    }
    helm_hip_program *prog = NULL;
    CHECK(helm_hip_program_create(ctx, op, i0, i1, i2, out, poff, n_launch, &prog), "helm_hip_program_create");

    /* evaluate_encrypted */
    CHECK(helm_hip_program_run(ctx, prog, wires, 0, n_launch), "helm_hip_program_run");
    CHECK(helm_hip_sync(ctx), "helm_hip_sync");

    /* evaluate_ready shape: one level of MUX(sel, new, old) - here sel = cout (true), new = s0, old = i1 (false) */
    const int32_t m_op[1] = {HELM_GATE_MUX}, m_i0[1] = {S0}, m_i1[1] = {I1}, m_i2[1] = {COUT}, m_out[1] = {T3};
    CHECK(helm_hip_eval_gate_level(ctx, wires, m_op, m_i0, m_i1, m_i2, m_out, 1), "helm_hip_eval_gate_level");

    /* decrypt_outputs (+ the internal wires the reference test checks, circuit_test.rs:37-44) */
    const int32_t want_rows[6] = {S0, S1, COUT, I0, I1, T3};
    const uint8_t want_bits[6] = {1, 1, 1, 0, 0, 1};
    uint32_t *dl = malloc(6 * row * 4);
    uint8_t got[6];
    CHECK(helm_hip_wires_download(ctx, wires, want_rows, dl, 6), "helm_hip_wires_download");
    CHECK(helm_client_decrypt_bool(ck, dl, 6, got), "decrypt");
    int bad = 0;
    for (int i = 0; i < 6; i++)
        if (got[i] != want_bits[i]) {
            fprintf(stderr, "wire row %d decrypts to %d, expected %d\n", want_rows[i], got[i], want_bits[i]);
            bad = 1;
        }

    /* Drop */
    CHECK(helm_hip_program_destroy(ctx, prog), "helm_hip_program_destroy");
    CHECK(helm_hip_wires_free(ctx, wires), "helm_hip_wires_free");
    CHECK(helm_hip_ctx_destroy(ctx), "helm_hip_ctx_destroy");
    helm_client_key_free(ck);
    free(t_bsk); free(t_ksk); free(bsk); free(ksk); free(cts); free(dl);
    if (bad) return 1;
    printf("ok: %s, %lld launch(es), 2-bit adder all-true -> sum[0]=sum[1]=cout=1, i0=i1=0\n", set, (long long)n_launch);
    return 0;
}
