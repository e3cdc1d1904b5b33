/*
 * helm_shortint.h — C ABI of the MI355X-native LUT-mode / arithmetic-mode engine
 * (64-bit torus, "shortint" ciphertexts: message + carry bits under one padding bit).
 *
 * Reference interfaces this replaces (all arithmetic inside the `tfhe` 0.4.1 crate,
 * Cargo.toml:18, absent from the tree):
 *   src/gates.rs:754-785      gates::lut(): tfhe::shortint::ServerKey::
 *                             {smart_evaluate_bivariate_function, smart_neg,
 *                              smart_scalar_left_shift, add, create_trivial,
 *                              generate_lookup_table, apply_lookup_table}
 *   src/circuit.rs:1032-1083  LutCircuit::evaluate_encrypted: per level,
 *                             gates.par_iter_mut() -> one lut() per gate
 *   src/circuit.rs:970-1000   encrypt_inputs (client_key.encrypt(u64), create_trivial(0))
 *   src/gates.rs:306-702      arithmetic-mode operators (FheUint8..128 = radix of shortint
 *                             blocks): served by the same two device primitives below
 *
 * Shape of the replacement.  Ciphertexts live in a device-resident table of "big" LWE
 * rows (k*N mask words + body, uint64) - tfhe's KS_PBS order: apply_lookup_table =
 * keyswitch big->small, then programmable bootstrap small->big.  Two level-batched
 * primitives carry every operator of both modes:
 *   helm_si_lincomb()      out = sum_t coef[t] * in[t] + const        (no bootstrap)
 *   helm_si_apply_luts()   out = PBS_lut( KS( in ) )                  (one bootstrap)
 * and helm_si_eval_lut_level() is gates::lut() for a whole netlist level.
 *
 * Conventions as in helm_hip.h: 0 / negative helm_status, helm_hip_last_error(),
 * caller owns host buffers, one context per device + host thread (several threads may share a
 * context if they serialise their calls - the host library's round merger does, under one lock),
 * stream-asynchronous with helm_si_sync(), no CPU fallback.
 *
 * Layouts (all words uint64_t, arithmetic mod 2^64)
 *   big LWE / wire row   k*N mask words + body
 *   bootstrapping key    [n][pbs_l][k+1][k+1][N]      (as helm_hip.h, 64-bit); multi-bit sets
 *                        (grouping_factor g > 1): [n/g][2^g][pbs_l][k+1][k+1][N]
 *   keyswitching key     [k*N][ks_l][n+1]
 *   encoding             value v in [0, message_modulus*carry_modulus) -> v * delta,
 *                        delta = 2^63 / (message_modulus * carry_modulus)
 *                        (the one citable line: src/gates.rs:851)
 *   look-up table        test polynomial of N words: box v of N/(msg*carry) coefficients
 *                        = f(v) * delta, rotated left by half a box, wrapped part negated
 */
#ifndef HELM_SHORTINT_H
#define HELM_SHORTINT_H

#include <stddef.h>
#include <stdint.h>
#include "helm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct helm_si_ctx helm_si_ctx;
typedef struct helm_si_wires helm_si_wires;

/* tfhe::shortint::{ClassicPBSParameters, MultiBitPBSParameters} as HELM picks them
 * (src/bin/helm.rs:301, tests/circuit_test.rs:287; src/bin/helm.rs:83 for arithmetic mode),
 * runtime values.
 * grouping_factor g: 0 or 1 = classical blind rotation (n CMUX steps, key = GGSW(s_i));
 * g = 2 or 3 = multi-bit blind rotation (tfhe's MultiBitPBS, reference helm.rs:83 uses g = 3):
 * n/g group steps; the key holds, per group, 2^g GGSWs of the indicators
 * prod_{i in S} s_i * prod_{i not in S} (1 - s_i), S a subset of the group (bit i of the subset
 * index = member i), and a step is acc <- (sum_S X^(sum_{i in S} a~_i) * GGSW_S) (x) acc.
 * n must be a multiple of g.
 * Accepted: pbs_logB <= 24 and pbs_logB * pbs_l <= 31 (every tfhe shortint set; the kernels multiply decomposition digits by
 * a 25-bit root of unity without a modular reduction), exact products below the CRT pair's range (2^97.5). */
typedef struct {
    int32_t n, k, N;
    int32_t pbs_l, pbs_logB;
    int32_t ks_l, ks_logB;
    int32_t message_modulus, carry_modulus;
    int32_t grouping_factor;
} helm_si_params;

/* Replaces shortint ServerKey construction (helm.rs:301: gen_keys(PARAM_...)). */
int helm_si_ctx_create(int device_id, const helm_si_params *params, helm_si_ctx **out);
int helm_si_ctx_destroy(helm_si_ctx *ctx);
/* A lane: a second context on the same device that shares `primary`'s keys and may work on its wire tables, with
 * its own stream and scratch - independent parts of a circuit can then be evaluated concurrently (one host thread
 * per lane; the GPU overlaps their launches).  The reference's unit of parallelism is the level
 * (src/circuit.rs:1057, 1321: par_iter over the gates of a level); lanes add parallelism ACROSS levels for
 * sub-circuits that share no wire.  Fork after the keys are loaded; destroy every lane before its primary.
 * Rows written through one lane must not be touched through another until both have been synchronised. */
int helm_si_ctx_fork(helm_si_ctx *primary, helm_si_ctx **lane_out);
int helm_si_get_params(const helm_si_ctx *ctx, helm_si_params *out);
/* The CRT pair of prime fields the bootstrap kernels of this context compute in, as its size class: 49 = 5072^4 + 1 and
 * 5096^4 + 1 (every parameter set), 46 = 2736^4 + 1 and 2872^4 + 1 - k > 1 contexts (k_pbs64k: the set reference
 * src/bin/helm.rs:301 installs for LUT mode) whose LOADED key keeps the exact products of a blind-rotation step below
 * p p' / 2 = 2^90.6: helm_si_load_bootstrap_key computes B/2 x the largest l1-norm of a key column for the key at hand (an
 * exact guarantee for that key and every input; a generated key fits, the worst case of the set does not and keeps 49), so
 * the value may change when a key is loaded (HELM_SI_FIELD=49 in the environment keeps the 49-bit pair).  Results do not
 * depend on it (exact integer arithmetic either way); reported because the operation count of the kernel does: the smaller
 * primes leave 2^53 / p >= 128 of headroom, so both leading stages of a forward transform on 17-bit digits are plain
 * multiplications and most recentrings go.  Negative on error. */
int helm_si_field_bits(const helm_si_ctx *ctx);
int helm_si_set_stream(helm_si_ctx *ctx, void *hip_stream);
int helm_si_sync(helm_si_ctx *ctx);
/* Dispatch priority of the context's OWN stream (no effect after helm_si_set_stream): high != 0 = the device's highest
 * stream priority, 0 = its lowest.  When two lanes have launches ready at the same time the workgroups of the
 * high-priority one are placed first - the host library gives it to the lane with the longest chain of bootstrap rounds.
 * Synchronises the stream it replaces. */
int helm_si_set_priority(helm_si_ctx *ctx, int high);
int helm_si_load_bootstrap_key(helm_si_ctx *ctx, const uint64_t *bsk_std, size_t n_words);
int helm_si_load_keyswitch_key(helm_si_ctx *ctx, const uint64_t *ksk, size_t n_words);

/* Device-resident ciphertext table (replaces HashMap<String, Arc<RwLock<CtxtShortInt>>>,
 * circuit.rs:1046-1049).  Rows of k*N+1 words. */
int helm_si_wires_alloc(helm_si_ctx *ctx, int64_t n_rows, helm_si_wires **out);
int helm_si_wires_free(helm_si_ctx *ctx, helm_si_wires *w);
int helm_si_wires_upload(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, const uint64_t *lwe_host,
                         int64_t count);
int helm_si_wires_download(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, uint64_t *lwe_host,
                           int64_t count);
/* Row copy between two tables of the same context (Ciphertext::clone, circuit.rs:1046-1049). */
int helm_si_wires_copy(helm_si_ctx *ctx, helm_si_wires *src, const int32_t *src_idx, helm_si_wires *dst,
                       const int32_t *dst_idx, int64_t count);
/* ServerKey::create_trivial(value) (circuit.rs:978): zero mask, body = value * delta. */
int helm_si_wires_set_trivial(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *idx, const uint64_t *value,
                              int64_t count);

/* out[g] = sum_{t<terms} coef[g*terms+t] * row in_idx[g*terms+t]  +  const_add[g] * delta
 * (in_idx = -1: term skipped).  Replaces unchecked add / sub / scalar_mul /
 * scalar_left_shift / neg / scalar_add of tfhe::shortint (gates.rs:769,776-778). */
int helm_si_lincomb(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int64_t *coef,
                    const int64_t *const_add, const int32_t *out_idx, int32_t terms, int64_t count);

/* Look-up tables as test polynomials (n_luts rows of N words), see layout above.
 * helm_si_make_lut() is ServerKey::generate_lookup_table(f) with f given as its value
 * table over [0, message_modulus*carry_modulus). */
int helm_si_make_lut(const helm_si_ctx *ctx, const uint64_t *f_values, uint64_t *test_poly_out);
/* out[g] = apply_lookup_table(row in_idx[g], luts[lut_idx[g]])  (gates.rs:783): keyswitch
 * then programmable bootstrap, `count` ciphertexts in one batched dispatch. */
int helm_si_apply_luts(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *in_idx, const int32_t *lut_idx,
                       const int32_t *out_idx, int64_t count, const uint64_t *luts, int64_t n_luts);

/* One netlist level of LUT gates = gates::lut() per gate (gates.rs:754-785):
 *   arity 1  : table all zero -> copy, else -> negation (smart_neg)
 *   arity 2  : bivariate f(x,y) = table[(x&1)*2 + (y&1)]
 *   arity >=3: pack sum in_i << (arity-1-i) (first input = MSB, gates.rs:159-167),
 *              f(x) = table[x] & 1
 * arity 0 with table 0: DFF / copy of in_idx[g*max_in] (circuit.rs:1063-1069).
 * table[g] holds the truth table as bits (bit i = entry i), in_idx is [count][max_in]. */
int helm_si_eval_lut_level(helm_si_ctx *ctx, helm_si_wires *w, const int32_t *arity, const int32_t *in_idx,
                           int32_t max_in, const uint64_t *table, const int32_t *out_idx, int64_t count);

/* Multi-GPU (one process per GPU, keys and wire tables replicated): after this call every
 * bootstrap batch of at least `min_batch` ciphertexts - helm_si_apply_luts() and everything
 * built on it: helm_si_eval_lut_level(), the radix operators of the host library - is split
 * into `world` contiguous chunks.  This rank keyswitches and bootstraps chunk `rank` into
 * `stage_dev` (rows of k*N+1 words), calls fn(user, rows_per_rank), which must all-gather
 * rows_per_rank rows of stage_dev from every rank into gather_dev in rank order ON THE
 * CONTEXT'S STREAM (ncclAllGather over RCCL/xGMI; helm_si_set_stream) and return 0, and then
 * scatters the gathered rows into the table.  Every rank must issue the same calls in the same
 * order; the ciphertexts are identical to a single-GPU evaluation.  stage_dev holds
 * capacity_rows rows, gather_dev capacity_rows * world (larger batches go in several rounds).
 * world < 1, or world = 1 without a callback, switches sharding off (world = 1 WITH a callback keeps
 * every batch on the stage -> collective -> scatter path: the single-GPU test of it).  The reference
 * has no multi-GPU path; its unit of parallelism is the level (src/circuit.rs:1057 par_iter_mut over
 * the gates of a level). */
typedef int (*helm_si_exchange_fn)(void *user, int64_t rows_per_rank);
int helm_si_set_exchange(helm_si_ctx *ctx, int32_t rank, int32_t world, int64_t min_batch, void *stage_dev,
                         void *gather_dev, int64_t capacity_rows, helm_si_exchange_fn fn, void *user);
/* The same with the collective INSIDE the library: the all-gather is ncclAllGather through `comm`
 * (include/helm_comm.h; rank and world are the communicator's) on the context's stream, into a gather
 * buffer of capacity_rows * world rows the context allocates; this rank's chunk is bootstrapped straight
 * into its slot of that buffer (all-gather in place).  No callback, no host framework - what a Rust host calls.
 * comm = NULL switches sharding off.  The communicator must outlive the setting. */
struct helm_comm;
int helm_si_set_exchange_comm(helm_si_ctx *ctx, struct helm_comm *comm, int64_t min_batch, int64_t capacity_rows);
/* batches sharded so far and rows moved through gather_dev (per rank) */
int helm_si_exchange_stats(const helm_si_ctx *ctx, int64_t *batches, int64_t *rows);
/* the `world` of helm_si_set_exchange (1: sharding off).  A lane does not inherit the exchange of its primary: callers
 * that shard keep to the primary context (the host library's ArithCircuit does not fork its default lane then). */
int helm_si_exchange_world(const helm_si_ctx *ctx);

/* Audit (tracing): while a callback is set, every helm_si_lincomb() and helm_si_apply_luts() call - and with them everything
 * built on the two: helm_si_eval_lut_level(), the LUT-mode and arithmetic-mode evaluators of the host library - copies its
 * operand rows (read BEFORE the call runs: a batch may work in place) and its result rows to the host and hands them to
 * `fn` together with the call's arguments; a non-zero return fails the call.  How tests/test_gpu_audit.py checks whole
 * evaluations at the full parameter sets against the CPU oracle, operation by operation (each batch's outputs == the
 * oracle's on the GPU's own inputs, hence every wire).  Slow by construction (two synchronous copies per call); lanes
 * forked AFTER this call inherit it.  fn = NULL switches it off. */
typedef struct {
    int32_t kind;             /* 0 = helm_si_apply_luts, 1 = helm_si_lincomb */
    int32_t terms;            /* lincomb: operands per output */
    int64_t count, n_luts;
    const uint64_t *in_rows;  /* apply_luts: count rows of k*N+1 words; lincomb: count * terms (a skipped operand: zeros) */
    const uint64_t *out_rows; /* count rows */
    const int32_t *lut_idx;   /* apply_luts */
    const uint64_t *luts;     /* apply_luts: n_luts test polynomials of N words */
    const int32_t *in_idx;    /* lincomb: the call's own arrays */
    const int64_t *coef, *const_add;
} helm_si_audit_record;
typedef int (*helm_si_audit_fn)(void *user, const helm_si_audit_record *rec);
int helm_si_set_audit(helm_si_ctx *ctx, helm_si_audit_fn fn, void *user);

/* Debug build only (-DHELM_CHECK_BOUNDS: csrc/libhelm_hip_check.so, loaded through HELM_HIP_LIB): the kernels of the 64-bit
 * engine (and of the WoP-PBS path, which shares its translation unit) count every violation of the contracts of the lazy
 * modular arithmetic, slots as in helm_hip_bound_violations (include/helm_hip.h).  The counters are per engine: the boolean
 * engine's are read through its own entry point.  The regular build returns HELM_ERR_STATE. */
int helm_si_bound_violations(helm_si_ctx *ctx, uint32_t counts[8], int reset);

/* Programmable bootstraps the device holds at once under this parameter set: CUs x workgroups of the set's bootstrap kernel
 * per CU (1 at N = 2048, 2 for k_pbs64k).  A batch of at most this many ciphertexts takes one bootstrap's time whatever its
 * size; the host library merges the look-up rounds of concurrent operators into launches of at most this size. */
int64_t helm_si_round_capacity(helm_si_ctx *ctx);

/* Primitive forms on host buffers (tests).  small: count x (n+1); big: count x (k*N+1). */
int helm_si_keyswitch_batch(helm_si_ctx *ctx, const uint64_t *in_big, uint64_t *out_small, int64_t count);
int helm_si_pbs_batch(helm_si_ctx *ctx, const uint64_t *in_small, const uint64_t *luts, int64_t n_luts,
                      const int32_t *lut_idx, uint64_t *out_big, int64_t count);

typedef struct {
    double pbs_ms, ks_ms, linear_ms;
    int64_t pbs_launches, pbs_count, ks_launches, ks_count;
} helm_si_timing;
int helm_si_timing_enable(helm_si_ctx *ctx, int enable);
int helm_si_get_timing(helm_si_ctx *ctx, helm_si_timing *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* HELM_SHORTINT_H */
