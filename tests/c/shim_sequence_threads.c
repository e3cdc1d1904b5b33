/*
 * shim_sequence_threads.c — a host that drives its ranks from THREADS of one process (one engine context per rank), in C,
 * holding nothing but the two shared libraries: include/helm_comm.h's helm_comm_create_in_process() gives every rank
 * thread a communicator whose all-gather is device-to-device copies between the ranks' buffers inside the library.
 *
 *   per rank thread:  helm_hip_ctx_create + keys, wire table with the SAME input ciphertexts, the 2-bit adder packed for
 *                     `world` ranks (helm_host_pack_levels_costed), helm_hip_program_run_sharded_comm (replicate_below = 0:
 *                     every launch that bootstraps is cut by bootstrap weight - levels of 1..4 gates over up to 8 ranks: most
 *                     ranks hold an empty, padded chunk - exchanged and scattered; every rank makes the same call, in order or
 *                     overlapped as the program argument says)
 *   main thread:      every rank's wire table == helm_hip_program_run on one context, word for word; the known answer of
 *                     reference tests/circuit_test.rs:17-45 decrypts.
 * The level of reference src/circuit.rs:531 is the sharded unit.
 * Usage: shim_sequence_threads [parameter set = toy_k2] [world = 8] [overlap = 0]
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "helm_client.h"
#include "helm_comm.h"
#include "helm_hip.h"
#include "helm_host.h"

enum { A0, A1, B0, B1, CIN, I0, S0, T0, T1, C1, I1, S1, T2, T3, COUT, N_WIRES };
static const int32_t OP[10] = {HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_XOR, HELM_GATE_AND,
                               HELM_GATE_OR, HELM_GATE_XOR, HELM_GATE_AND, HELM_GATE_OR};
static const int32_t IN0[10] = {A0, A0, A1, A1, I0, I0, T0, I1, I1, T2};
static const int32_t IN1[10] = {B0, B0, B1, B1, CIN, CIN, T1, C1, C1, T3};
static const int32_t IN2[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static const int32_t OUT[10] = {I0, T0, I1, T2, S0, T1, C1, S1, T3, COUT};
#define MAX_WORLD 16

typedef struct {
    int rank, world, overlap, failed;
    helm_hip_params P;
    helm_client_key *ck;
    const uint32_t *cts; /* the five input ciphertexts, shared by every rank */
    helm_comm *comm;
    uint32_t *table;     /* N_WIRES rows, filled by the rank */
    char err[512];
} rank_t;

#define RCHECK(call, what)                                                                                     \
    do {                                                                                                       \
        int rc__ = (call);                                                                                     \
        if (rc__ != 0) {                                                                                       \
            snprintf(R->err, sizeof(R->err), "rank %d: %s failed (%d): hip='%s' host='%s'", R->rank, what, rc__, \
                     helm_hip_last_error(), helm_host_last_error());                                           \
            R->failed = 1;                                                                                     \
            helm_comm_abort_group(R->comm); /* nobody waits for a rank that has left */                        \
            return NULL;                                                                                       \
        }                                                                                                      \
    } while (0)

static void *rank_main(void *arg)
{
    rank_t *R = arg;
    helm_hip_ctx *ctx = NULL;
    RCHECK(helm_hip_ctx_create(0, &R->P, &ctx), "helm_hip_ctx_create");
    RCHECK(helm_hip_load_bootstrap_key(ctx, helm_client_bsk(R->ck), helm_client_bsk_words(R->ck)), "load_bootstrap_key");
    RCHECK(helm_hip_load_keyswitch_key(ctx, helm_client_ksk(R->ck), helm_client_ksk_words(R->ck)), "load_keyswitch_key");
    const int32_t in_rows[5] = {A0, A1, B0, B1, CIN};
    uint8_t zeros[10] = {0};
    helm_hip_wires *w = NULL;
    RCHECK(helm_hip_wires_alloc(ctx, N_WIRES, &w), "wires_alloc");
    RCHECK(helm_hip_wires_set_trivial(ctx, w, OUT, zeros, 10), "set_trivial");
    RCHECK(helm_hip_wires_upload(ctx, w, in_rows, R->cts, 5), "wires_upload");
    const int64_t off[6] = {0, 4, 6, 7, 9, 10};
    int32_t op[10], i0[10], i1[10], i2[10], out[10];
    int64_t order[10], poff[11], n_launch = 0;
    double cost[4];
    RCHECK(helm_hip_launch_costs(ctx, cost), "launch_costs");
    const int64_t q = helm_hip_launch_quantum(ctx);
    RCHECK(q <= 0 || helm_host_pack_levels_costed(OP, IN0, IN1, IN2, OUT, off, 5, q * R->world, cost, order, poff, &n_launch) < 0,
           "pack_levels_costed");
    for (int g = 0; g < 10; g++) {
        op[g] = OP[order[g]]; i0[g] = IN0[order[g]]; i1[g] = IN1[order[g]]; i2[g] = IN2[order[g]]; out[g] = OUT[order[g]];
    }
    helm_hip_program *prog = NULL;
    RCHECK(helm_hip_program_create(ctx, op, i0, i1, i2, out, poff, n_launch, &prog), "program_create");
    if (R->rank == R->world - 1) { /* the cut the engine makes, seen through the ABI: rank r owns gates bounds[r] .. bounds[r + 1] */
        int64_t b[MAX_WORLD + 1];
        RCHECK(helm_hip_program_chunk_bounds(prog, 0, R->world, b), "chunk_bounds");
        RCHECK(!(b[0] == 0 && b[R->world] == poff[1] - poff[0]), "chunk_bounds: ends");
    }
    RCHECK(helm_hip_program_run_sharded_comm(ctx, prog, w, R->comm, 0, R->overlap), "program_run_sharded_comm");
    RCHECK(helm_hip_sync(ctx), "sync");
    int32_t all[N_WIRES];
    for (int i = 0; i < N_WIRES; i++) all[i] = i;
    RCHECK(helm_hip_wires_download(ctx, w, all, R->table, N_WIRES), "wires_download");
    double v = (double)(R->rank + 1); /* the host-side helper through the same group: sum over the ranks */
    RCHECK(helm_comm_all_reduce_f64(R->comm, &v, 0), "all_reduce_f64");
    RCHECK(v != (double)R->world * (R->world + 1) / 2, "all_reduce_f64: value");
    RCHECK(helm_hip_program_destroy(ctx, prog), "program_destroy");
    RCHECK(helm_hip_wires_free(ctx, w), "wires_free");
    RCHECK(helm_hip_ctx_destroy(ctx), "ctx_destroy");
    return NULL;
}

int main(int argc, char **argv)
{
    const char *set = argc > 1 ? argv[1] : "toy_k2";
    const int world = argc > 2 ? atoi(argv[2]) : 8, overlap = argc > 3 ? atoi(argv[3]) : 0;
    if (world < 1 || world > MAX_WORLD) return 2;
    helm_hip_params P;
    double lwe_std, glwe_std;
    if (helm_client_named_params(set, &P, &lwe_std, &glwe_std)) { fprintf(stderr, "named_params: %s\n", helm_client_last_error()); return 1; }
    helm_client_key *ck = NULL;
    if (helm_client_keygen(&P, lwe_std, glwe_std, 7, &ck)) { fprintf(stderr, "keygen: %s\n", helm_client_last_error()); return 1; }
    const size_t row = (size_t)P.n + 1;
    const uint8_t in_bits[5] = {1, 1, 1, 1, 1};
    uint32_t *cts = malloc(5 * row * 4);
    if (helm_client_encrypt_bool(ck, in_bits, 5, cts)) return 1;

    /* the one-context reference pass (also the first context of the process) */
    helm_hip_ctx *ctx = NULL;
    if (helm_hip_ctx_create(0, &P, &ctx)) { fprintf(stderr, "helm_hip_ctx_create failed: %s\n", helm_hip_last_error()); return 1; }
    if (helm_hip_load_bootstrap_key(ctx, helm_client_bsk(ck), helm_client_bsk_words(ck)) ||
        helm_hip_load_keyswitch_key(ctx, helm_client_ksk(ck), helm_client_ksk_words(ck))) return 1;
    const int32_t in_rows[5] = {A0, A1, B0, B1, CIN};
    uint8_t zeros[10] = {0};
    helm_hip_wires *w = NULL;
    const int64_t off[6] = {0, 4, 6, 7, 9, 10};
    helm_hip_program *prog = NULL;
    uint32_t *want = malloc(N_WIRES * row * 4);
    int32_t all[N_WIRES];
    for (int i = 0; i < N_WIRES; i++) all[i] = i;
    if (helm_hip_wires_alloc(ctx, N_WIRES, &w) || helm_hip_wires_set_trivial(ctx, w, OUT, zeros, 10) ||
        helm_hip_wires_upload(ctx, w, in_rows, cts, 5) || helm_hip_program_create(ctx, OP, IN0, IN1, IN2, OUT, off, 5, &prog) ||
        helm_hip_program_run(ctx, prog, w, 0, 5) || helm_hip_wires_download(ctx, w, all, want, N_WIRES)) {
        fprintf(stderr, "reference pass: %s\n", helm_hip_last_error());
        return 1;
    }
    helm_hip_program_destroy(ctx, prog);
    helm_hip_wires_free(ctx, w);
    helm_hip_ctx_destroy(ctx);

    int devices[MAX_WORLD] = {0};
    helm_comm *comms[MAX_WORLD];
    if (helm_comm_create_in_process(devices, world, 120.0, comms)) { fprintf(stderr, "helm_comm_create_in_process: %s\n", helm_hip_last_error()); return 1; }
    rank_t R[MAX_WORLD];
    pthread_t th[MAX_WORLD];
    for (int r = 0; r < world; r++) {
        memset(&R[r], 0, sizeof(R[r]));
        R[r].rank = r; R[r].world = world; R[r].overlap = overlap; R[r].P = P; R[r].ck = ck; R[r].cts = cts; R[r].comm = comms[r];
        R[r].table = malloc(N_WIRES * row * 4);
        if (pthread_create(&th[r], NULL, rank_main, &R[r])) return 1;
    }
    int bad = 0;
    for (int r = 0; r < world; r++) {
        pthread_join(th[r], NULL);
        if (R[r].failed) { fprintf(stderr, "%s\n", R[r].err); bad = 1; }
    }
    if (bad) return 1;
    for (int r = 0; r < world; r++) {
        int rr = -1, ww = -1, dd = -1, ver = -1;
        if (helm_comm_info(comms[r], &rr, &ww, &dd, &ver) || rr != r || ww != world || ver != 0) { fprintf(stderr, "comm_info of rank %d\n", r); return 1; }
        if (memcmp(R[r].table, want, N_WIRES * row * 4) != 0) { fprintf(stderr, "rank %d: the sharded pass differs from helm_hip_program_run\n", r); return 1; }
        helm_comm_destroy(comms[r]);
        free(R[r].table);
    }
    const int32_t want_rows[5] = {S0, S1, COUT, I0, I1};
    const uint8_t want_bits[5] = {1, 1, 1, 0, 0};
    uint8_t got[N_WIRES];
    if (helm_client_decrypt_bool(ck, want, N_WIRES, got)) return 1;
    for (int i = 0; i < 5; i++)
        if (got[want_rows[i]] != want_bits[i]) { fprintf(stderr, "wire row %d decrypts to %d\n", want_rows[i], got[want_rows[i]]); return 1; }
    helm_client_key_free(ck);
    free(cts); free(want);
    printf("ok: %s, %d rank threads over helm_comm_create_in_process (%s exchange): every rank's wire table identical to the one-context pass\n",
           set, world, overlap ? "overlapped" : "in-order");
    return 0;
}
