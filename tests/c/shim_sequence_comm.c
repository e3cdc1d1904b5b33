/*
 * shim_sequence_comm.c — the multi-GPU call sequence of rust/helm-hip/src/multi_gpu.rs, from C, in a process that holds
 * NOTHING but the two shared libraries: no Python, no torch, no RCCL binding of its own.
 *
 *   HipComm::unique_id / HipComm::new   -> helm_comm_get_unique_id / helm_comm_create  (ncclCommInitRank inside the library;
 *                                          RCCL bound with dlopen from the loader's search path / /opt/rocm)
 *   HipGateCircuit::shard_over + build_program with the world size
 *                                        -> helm_hip_launch_costs, helm_host_pack_levels_costed(quantum = world x round)
 *   evaluate_encrypted                  -> helm_hip_program_run_sharded_comm (replicate_below = 0: every launch that
 *                                          bootstraps goes stage -> in-place ncclAllGather -> scatter)
 *
 * One process = one rank; `world` = 1 here (a one-GPU box; RCCL refuses two ranks on one device).  The sharded pass must
 * leave the wire table helm_hip_program_run leaves, word for word, and the 2-bit adder of reference
 * tests/circuit_test.rs:17-45 must decrypt (all inputs true => sum[0] = sum[1] = cout = 1, i0 = i1 = 0).
 * The level of reference src/circuit.rs:531 is the sharded unit.
 * Usage: shim_sequence_comm [parameter set name]   (default boolean_default)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "helm_client.h"
#include "helm_comm.h"
#include "helm_hip.h"
#include "helm_host.h"

#define CHECK(call, what)                                                                       \
    do {                                                                                        \
        int rc__ = (call);                                                                      \
        if (rc__ != 0) {                                                                        \
            fprintf(stderr, "%s failed (%d): hip='%s' client='%s' host='%s'\n", what, rc__,     \
                    helm_hip_last_error(), helm_client_last_error(), helm_host_last_error());   \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

enum { A0, A1, B0, B1, CIN, I0, S0, T0, T1, C1, I1, S1, T2, T3, COUT, N_WIRES };
static const int32_t OP[10] = {HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_XOR, HELM_GATE_AND,
                               HELM_GATE_OR, HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_OR};
static const int32_t IN0[10] = {A0, A0, A1, A1, I0, I0, T0, I1, I1, T2};
static const int32_t IN1[10] = {B0, B0, B1, B1, CIN, CIN, T1, C1, C1, T3};
static const int32_t IN2[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static const int32_t OUT[10] = {I0, T0, I1, T2, S0, T1, C1, S1, T3, COUT};

int main(int argc, char **argv)
{
    const char *set = argc > 1 ? argv[1] : "boolean_default";
    const int rank = 0, world = 1, device = 0;
    helm_hip_params P;
    double lwe_std, glwe_std;
    CHECK(helm_client_named_params(set, &P, &lwe_std, &glwe_std), "named_params");
    helm_client_key *ck = NULL;
    CHECK(helm_client_keygen(&P, lwe_std, glwe_std, 7 /* the deterministic test generator: every rank draws the same keys */, &ck),
          "keygen");
    helm_hip_ctx *ctx = NULL;
    CHECK(helm_hip_ctx_create(device, &P, &ctx), "helm_hip_ctx_create");
    CHECK(helm_hip_load_bootstrap_key(ctx, helm_client_bsk(ck), helm_client_bsk_words(ck)), "helm_hip_load_bootstrap_key");
    CHECK(helm_hip_load_keyswitch_key(ctx, helm_client_ksk(ck), helm_client_ksk_words(ck)), "helm_hip_load_keyswitch_key");

    /* HipComm: rank 0 draws the id (and would ship it to the other processes), every rank joins */
    if (!helm_comm_available()) {
        fprintf(stderr, "no RCCL library could be bound: %s\n", helm_hip_last_error());
        return 1;
    }
    uint8_t id[HELM_COMM_ID_BYTES];
    CHECK(helm_comm_get_unique_id(id), "helm_comm_get_unique_id");
    helm_comm *comm = NULL;
    CHECK(helm_comm_create(device, id, rank, world, &comm), "helm_comm_create");
    int r = -1, w = -1, d = -1, version = 0;
    CHECK(helm_comm_info(comm, &r, &w, &d, &version), "helm_comm_info");
    if (r != rank || w != world || d != device) {
        fprintf(stderr, "RCCL reports rank %d of %d on device %d\n", r, w, d);
        return 1;
    }

    /* two wire tables with the same inputs: one for the sharded pass, one for helm_hip_program_run */
    const size_t row = (size_t)P.n + 1;
    const int32_t in_rows[5] = {A0, A1, B0, B1, CIN};
    const uint8_t in_bits[5] = {1, 1, 1, 1, 1};
    uint32_t *cts = malloc(5 * row * 4);
    CHECK(helm_client_encrypt_bool(ck, in_bits, 5, cts), "encrypt");
    helm_hip_wires *wires[2] = {NULL, NULL};
    uint8_t zeros[10] = {0};
    for (int t = 0; t < 2; t++) {
        CHECK(helm_hip_wires_alloc(ctx, N_WIRES, &wires[t]), "helm_hip_wires_alloc");
        CHECK(helm_hip_wires_set_trivial(ctx, wires[t], OUT, zeros, 10), "helm_hip_wires_set_trivial");
        CHECK(helm_hip_wires_upload(ctx, wires[t], in_rows, cts, 5), "helm_hip_wires_upload");
    }

    /* build_program for `world` ranks: a round per rank, the engine's cost per launch width */
    const int64_t off[6] = {0, 4, 6, 7, 9, 10};
    int32_t op[10], i0[10], i1[10], i2[10], out[10];
    int64_t order[10], poff[11], n_launch = 0;
    double cost[4];
    CHECK(helm_hip_launch_costs(ctx, cost), "helm_hip_launch_costs");
    const int64_t q = helm_hip_launch_quantum(ctx);
    if (q <= 0 || helm_host_pack_levels_costed(OP, IN0, IN1, IN2, OUT, off, 5, q * world, cost, order, poff, &n_launch) < 0) {
        fprintf(stderr, "pack_levels_costed: %s / %s\n", helm_host_last_error(), helm_hip_last_error());
        return 1;
    }
    for (int g = 0; g < 10; g++) {
        op[g] = OP[order[g]]; i0[g] = IN0[order[g]]; i1[g] = IN1[order[g]]; i2[g] = IN2[order[g]]; out[g] = OUT[order[g]];
    }
    helm_hip_program *prog = NULL;
    CHECK(helm_hip_program_create(ctx, op, i0, i1, i2, out, poff, n_launch, &prog), "helm_hip_program_create");

    /* evaluate_encrypted, sharded: every launch through the communicator; and the one-GPU pass next to it */
    CHECK(helm_hip_timing_enable(ctx, 1), "helm_hip_timing_enable");
    CHECK(helm_hip_program_run_sharded_comm(ctx, prog, wires[0], comm, 0, 0), "helm_hip_program_run_sharded_comm");
    CHECK(helm_hip_program_run(ctx, prog, wires[1], 0, n_launch), "helm_hip_program_run");
    CHECK(helm_hip_sync(ctx), "helm_hip_sync");
    helm_hip_timing tm;
    CHECK(helm_hip_get_timing(ctx, &tm, 1), "helm_hip_get_timing");
    int64_t collectives = 0, bytes = 0;
    CHECK(helm_comm_stats(comm, &collectives, &bytes), "helm_comm_stats");
    if (tm.exchange_count != n_launch || collectives != n_launch || bytes != tm.exchange_bytes || bytes != (int64_t)(10 * row * 4)) {
        fprintf(stderr, "%lld launches, %lld exchanges timed, %lld collectives, %lld / %lld bytes\n", (long long)n_launch,
                (long long)tm.exchange_count, (long long)collectives, (long long)bytes, (long long)tm.exchange_bytes);
        return 1;
    }

    int32_t all[N_WIRES];
    for (int i = 0; i < N_WIRES; i++) all[i] = i;
    uint32_t *t0 = malloc(N_WIRES * row * 4), *t1 = malloc(N_WIRES * row * 4);
    CHECK(helm_hip_wires_download(ctx, wires[0], all, t0, N_WIRES), "helm_hip_wires_download");
    CHECK(helm_hip_wires_download(ctx, wires[1], all, t1, N_WIRES), "helm_hip_wires_download");
    if (memcmp(t0, t1, N_WIRES * row * 4) != 0) {
        fprintf(stderr, "the sharded pass differs from helm_hip_program_run\n");
        return 1;
    }
    const int32_t want_rows[5] = {S0, S1, COUT, I0, I1};
    const uint8_t want_bits[5] = {1, 1, 1, 0, 0};
    uint8_t got[N_WIRES];
    CHECK(helm_client_decrypt_bool(ck, t0, N_WIRES, got), "decrypt");
    for (int i = 0; i < 5; i++)
        if (got[want_rows[i]] != want_bits[i]) {
            fprintf(stderr, "wire row %d decrypts to %d, expected %d\n", want_rows[i], got[want_rows[i]], want_bits[i]);
            return 1;
        }
    /* the control-plane helpers a host without another one uses */
    double v = 3.5;
    CHECK(helm_comm_all_reduce_f64(comm, &v, 1), "helm_comm_all_reduce_f64");
    CHECK(helm_comm_barrier(comm), "helm_comm_barrier");
    if (v != 3.5) return 1;

    CHECK(helm_hip_program_destroy(ctx, prog), "helm_hip_program_destroy");
    for (int t = 0; t < 2; t++) CHECK(helm_hip_wires_free(ctx, wires[t]), "helm_hip_wires_free");
    CHECK(helm_comm_destroy(comm), "helm_comm_destroy");
    CHECK(helm_hip_ctx_destroy(ctx), "helm_hip_ctx_destroy");
    helm_client_key_free(ck);
    free(cts); free(t0); free(t1);
    printf("ok: %s, RCCL %d, %lld launch(es) each through ncclAllGather inside the library, wire table identical to the one-GPU pass\n",
           set, version, (long long)n_launch);
    return 0;
}
