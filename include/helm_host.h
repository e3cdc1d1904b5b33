/*
 * helm_host.h — flat C view of the C++ host front end (helm_amd/csrc/host/helm_host.hpp)
 * for language bindings (the Python test harness binds it with ctypes).
 *
 * Every call mirrors one item of the reference's Rust API; names follow it:
 *   verilog_parser::{read_verilog_file, read_input_wires, write_output_wires}
 *                                          reference src/verilog_parser.rs:138-349
 *   parse_input_wire / hex_to_bitstring    reference src/lib.rs:90-106, 181-194
 *   Circuit::{new, sort_circuit, compute_levels, evaluate, ...}
 *                                          reference src/circuit.rs:104-381
 *   GateCircuit + trait EvalCircuit        reference src/circuit.rs:35-58, 449-577
 *
 * Conventions: 0 = ok, -1 = the reference would have panicked (message from
 * helm_host_last_error()).  Returned `char*` texts are malloc'd; free them with
 * helm_host_free().  Lists are newline-separated; wire maps are lines of
 * "name<TAB>Kind<TAB>value" with Kind in None/Bool/U8/U16/U32/U64/U128.
 */
#ifndef HELM_HOST_H
#define HELM_HOST_H
#include <stdint.h>
#include "helm_client.h"
#include "helm_hip.h"
#include "helm_shortint.h"
#include "helm_wopbs.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct helm_netlist helm_netlist;          /* parsed netlist: (gates, wire_set, inputs, outputs, dff_outputs, has_luts, has_arith) */
typedef struct helm_circuit helm_circuit;          /* Circuit */
typedef struct helm_gate_circuit helm_gate_circuit; /* GateCircuit */
typedef struct helm_enc_map helm_enc_map;          /* HashMap<String, Ciphertext> on the device */

const char *helm_host_last_error(void);
void helm_host_free(char *text);

/* verilog_parser */
int helm_host_read_verilog_file(const char *file_name, int is_arith, helm_netlist **out);
int helm_host_read_verilog_text(const char *text, int is_arith, helm_netlist **out);
void helm_host_netlist_free(helm_netlist *nl);
/* which: 0 gates (one line per gate: name<TAB>Type<TAB>output<TAB>lut_const_or_-<TAB>in0,in1,...),
 *        1 wire_set, 2 inputs, 3 outputs, 4 dff_outputs */
char *helm_host_netlist_list(const helm_netlist *nl, int which);
int helm_host_netlist_flags(const helm_netlist *nl, int *has_luts, int *has_arith);
int helm_host_read_input_wires(const char *file_name, const char *ptxt_type, char **out_map);
int helm_host_write_output_wires(const char *file_name, const char *wire_map);
int helm_host_parse_input_wire(const char *wire, const char *ptxt_type, char **out_value);
int helm_host_hex_to_bitstring(const char *hex, char **out_bits);

/* Circuit */
int helm_host_circuit_new(const helm_netlist *gates_from, const char *input_wires, const char *output_wires,
                          const char *dff_outputs, helm_circuit **out);
void helm_host_circuit_free(helm_circuit *c);
int helm_host_circuit_sort_circuit(helm_circuit *c);
int helm_host_circuit_compute_levels(helm_circuit *c);
/* gates (same line format as netlist_list 0, plus <TAB>level) in ordered_gates order */
char *helm_host_circuit_get_ordered_gates(const helm_circuit *c);
/* level_map: same gate lines, ascending level */
char *helm_host_circuit_level_map(const helm_circuit *c);
int helm_host_circuit_initialize_wire_map(const helm_circuit *c, const char *wire_set, const char *user_inputs,
                                          const char *ptxt_type, char **out_map);
int helm_host_circuit_evaluate(helm_circuit *c, const char *wire_map, char **out_map);

/* The benchmark suite's `preprocessor` binary (reference README.md:116-120,133-137; its source lives in an
 * un-vendored submodule): raw Yosys structural Verilog - or, with `arithmetic`, behavioural assign statements
 * over + - * / << >> - to the dialect helm_host_read_verilog_* reads.  *out is malloc'd text. */
int helm_host_preprocess(const char *text, int arithmetic, char **out);

/* Launch packing of a level schedule (the level loop of circuit.rs:524-543 made GPU-shaped): gates in level
 * order as index arrays (helm_hip_program_create's arguments) -> `order[total]` (new position -> gate index)
 * and `new_offsets` (room for total + 1 entries; *n_launches + 1 are written): dependency order is kept, a
 * launch holds a whole number of `quantum` bootstraps (helm_hip_launch_quantum()) while that many gates are
 * ready.  Returns 0 packed, 1 schedule kept unchanged (state-writing gates feed later gates), -1 error. */
int helm_host_pack_levels(const int32_t *opcode, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                          const int32_t *out, const int64_t *level_offsets, int64_t n_levels, int64_t quantum,
                          int64_t *order, int64_t *new_offsets, int64_t *n_launches);

/* The same with the engine's cost table: quarter_cost[q] = cost of a launch of at most (q + 1) / 4 of `quantum`
 * bootstraps relative to a full round (helm_hip_launch_costs(); NULL = helm_host_pack_levels).  A launch narrower than
 * a round then takes the width with the best bootstraps-per-cost among {everything ready, the quarter steps below it}
 * and leaves the rest for the next launch while gates are still waiting for their producers.  Same dependency order,
 * same ciphertexts; what changes is where the launch boundaries fall - this is what keeps per-rank chunks of a
 * launch sharded over N GPUs (quantum = N x helm_hip_launch_quantum()) on the engine's efficient widths. */
int helm_host_pack_levels_costed(const int32_t *opcode, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                 const int32_t *out, const int64_t *level_offsets, int64_t n_levels, int64_t quantum,
                                 const double *quarter_cost, int64_t *order, int64_t *new_offsets, int64_t *n_launches);

/* The engine's cut of one launch into `world` contiguous chunks, one per GPU (helm_amd/csrc/shard_rule.h; what
 * helm_hip_program_chunk_bounds() returns for an uploaded program): by BOOTSTRAP weight - binary gate 1, MUX 2, NOT / BUF /
 * DFF / constants 0 - so that a sharded launch gives every rank the same number of bootstraps to within one gate
 * (the reference balances the level dynamically, circuit.rs:531 `par_iter_mut`).  bounds has world + 1 entries;
 * returns the largest chunk (rows of one rank's slot in the all-gather), -1 on bad arguments. */
int64_t helm_host_shard_bounds(const int32_t *opcode, int64_t count, int world, int64_t *bounds);

/* encrypted wire maps */
int helm_host_enc_map_new(helm_hip_ctx *server_key, helm_enc_map **out);
void helm_host_enc_map_free(helm_enc_map *m);
int helm_host_enc_map_insert(helm_enc_map *m, const char *wire, const uint32_t *lwe);
int helm_host_enc_map_get(const helm_enc_map *m, const char *wire, uint32_t *lwe_out);
int helm_host_enc_map_contains_key(const helm_enc_map *m, const char *wire);
char *helm_host_enc_map_keys(const helm_enc_map *m);

/* GateCircuit::new(client_key, server_key, circuit) — the circuit is copied */
int helm_host_gate_circuit_new(helm_client_key *client_key, helm_hip_ctx *server_key, const helm_circuit *circuit,
                               helm_gate_circuit **out);
void helm_host_gate_circuit_free(helm_gate_circuit *gc);
int helm_host_gate_circuit_encrypt_inputs(helm_gate_circuit *gc, const char *wire_set, const char *input_wire_map,
                                          helm_enc_map **out);
int helm_host_gate_circuit_evaluate_encrypted(helm_gate_circuit *gc, const helm_enc_map *enc_wire_map,
                                              int64_t current_cycle, const char *ptxt_type, helm_enc_map **out);
int helm_host_gate_circuit_init_ready(helm_gate_circuit *gc, helm_enc_map **out);
int helm_host_gate_circuit_evaluate_ready(helm_gate_circuit *gc, const helm_enc_map *enc_wire_map,
                                          helm_enc_map *valid_outputs);
int helm_host_gate_circuit_decrypt_outputs(helm_gate_circuit *gc, const helm_enc_map *enc_wire_map, int verbose,
                                           char **out_map);
/* progress / output lines the reference prints (circuit.rs:542, 562-573); drains the buffer */
char *helm_host_gate_circuit_log(helm_gate_circuit *gc);
int64_t helm_host_gate_circuit_pbs_per_cycle(const helm_gate_circuit *gc);
/* Same-cycle memo (reference src/gates.rs:55-59: a Gate keeps `cycle` and its last encrypted output): evaluate_encrypted
 * called again with the same cycle on the very same, unmodified map returns the cached wire map without a launch; this
 * counts those calls.  (The reference's boolean cache probe is commented out, gates.rs:247-252, so the memo is only
 * allowed where it cannot be observed: identical inputs.) */
int64_t helm_host_gate_circuit_memo_hits(const helm_gate_circuit *gc);
/* Multi-GPU, one process per GPU: from the next evaluate_encrypted on the launches are packed for the world size of
 * `comm` (include/helm_comm.h: the engine's own RCCL communicator) and every launch of more than `replicate_below`
 * bootstraps is split over its ranks - the level of reference src/circuit.rs:531 is the sharded unit - the output
 * ciphertexts all-gathered with ncclAllGather inside the engine (helm_hip_program_run_sharded_comm).  Keys, circuit and
 * input ciphertexts must be the same on every rank; every rank gets the wire map of a one-GPU evaluation.
 * comm = NULL: back to one GPU. */
struct helm_comm;
int helm_host_gate_circuit_shard_over(helm_gate_circuit *gc, struct helm_comm *comm, int64_t replicate_below);
/* on != 0: a launch's all-gather + scatter run on the engine's exchange stream while the launches that do not need its
 * outputs go on (helm_hip_program_run_sharded_comm, overlap = 1); same wire map. */
int helm_host_gate_circuit_set_exchange_overlap(helm_gate_circuit *gc, int on);

/* ---- LUT mode / arithmetic mode (include/helm_shortint.h) ------------------------------
 * LutCircuit (circuit.rs:969-1120) and ArithCircuit (circuit.rs:1112-1500); `mode` 0 = LUT,
 * 1 = arithmetic.  Encrypted maps hold `blocks` big-LWE rows per wire (1 for LUT mode). */
typedef struct helm_si_circuit helm_si_circuit;
typedef struct helm_si_enc_map helm_si_enc_map;
/* client_key may be NULL: an evaluation-only circuit for hosts that encrypt and decrypt with their own keys (tfhe's, in
 * the Rust shim) - inputs go in through helm_host_si_enc_map_new / _insert, results come out through _get, and
 * evaluate_encrypted brings everything this library adds to a whole circuit (merged rounds of independent
 * sub-circuits, carry-save products and sums: DESIGN.md section 5) that the level-wise helm_host_radix_level cannot.
 * encrypt_inputs / decrypt_outputs then fail with a message. */
int helm_host_si_circuit_new(int mode, helm_si_client_key *client_key, helm_si_ctx *server_key,
                             const helm_circuit *circuit, helm_si_circuit **out);
void helm_host_si_circuit_free(helm_si_circuit *c);
int helm_host_si_circuit_encrypt_inputs(helm_si_circuit *c, const char *wire_set, const char *input_wire_map,
                                        helm_si_enc_map **out);
int helm_host_si_circuit_evaluate_encrypted(helm_si_circuit *c, const helm_si_enc_map *enc_wire_map,
                                            int64_t current_cycle, const char *ptxt_type, helm_si_enc_map **out);
int helm_host_si_circuit_init_ready(helm_si_circuit *c, helm_si_enc_map **out);
int helm_host_si_circuit_evaluate_ready(helm_si_circuit *c, const helm_si_enc_map *enc_wire_map,
                                        helm_si_enc_map *valid_outputs);
int helm_host_si_circuit_decrypt_outputs(helm_si_circuit *c, const helm_si_enc_map *enc_wire_map, int verbose,
                                         char **out_map);
/* LUT mode: gates with more inputs than one block holds (log2(message_modulus * carry_modulus) index bits) go
 * through the WoP-PBS path of include/helm_wopbs.h - Gate::evaluate_encrypted_high_precision_lut (gates.rs:721-742),
 * which the reference defines but never calls.  wop = NULL switches it off again. */
int helm_host_si_circuit_set_wopbs(helm_si_circuit *c, helm_wop_ctx *wop, int bits_per_block);
/* Arithmetic mode: one more lane (helm_si_ctx_fork of the circuit's server key).  Sub-circuits that share no wire are
 * then evaluated concurrently, one lane each, instead of meeting at every level boundary; identical ciphertexts.
 * lane = NULL removes all lanes.  helm_host_si_circuit_pbs_rounds_per_cycle() then reports the longest lane. */
int helm_host_si_circuit_add_lane(helm_si_circuit *c, helm_si_ctx *lane);
/* Arithmetic mode: carry-save products (default on).  A product whose every consumer is an addition or a subtraction
 * (possibly behind multiplications by powers of four) hands over the two terms its reduction ends with; the consumer
 * sums the terms of both operands and propagates carries once - a * b - c * d costs 11 + 7 rounds of bootstraps in a row
 * instead of 11 + 6 + 6.  Same values mod 2^bits on every wire; 0 = every operator propagates (as round 2). */
int helm_host_si_circuit_set_lazy_carries(helm_si_circuit *c, int on);
/* Arithmetic mode, merged rounds: launches of at most `capacity` ciphertexts (0 = helm_si_round_capacity(), the default).
 * A round that does not fit is cut; the cut is safe because every round lists the readers of a row before its in-place
 * writer - checked for every round, a violation is an error whatever the capacity.  Tests force capacity 1. */
int helm_host_si_circuit_set_round_capacity(helm_si_circuit *c, int64_t capacity);
/* Arithmetic mode, same-cycle memo (see helm_host_si_circuit_memo_hits): on by default and keyed on the cycle ALONE as the
 * reference's (src/gates.rs:307-312) - evaluate_encrypted with an already evaluated cycle returns that cycle's gate outputs
 * WHATEVER the inputs.  set_memo(0) switches it off, reset_memo() forgets the remembered cycle; changing lanes, lazy carries
 * or the round capacity resets it too. */
int helm_host_si_circuit_set_memo(helm_si_circuit *c, int on);
int helm_host_si_circuit_reset_memo(helm_si_circuit *c);
/* LUT mode: the per-gate `PBS time: {} us` lines of src/gates.rs:293-302 (default on) cost one host synchronisation per
 * level; 0 drops the lines and the synchronisation. */
int helm_host_si_circuit_set_timing_lines(helm_si_circuit *c, int on);
/* One level of arithmetic-mode operators as ONE batched call on a table of radix integers - what the reference does
 * gate by gate with the FheUintN operators (src/gates.rs:306-702: evaluate_encrypted_{copy,mul,div,add,sub,shift}_block
 * and their _plain forms; level loop src/circuit.rs:1320-1441).  An integer is `blocks` consecutive rows (2 message bits
 * per block, least significant first; 4..64 blocks for FheUint8..128); a, b, out are FIRST rows; an operator with a
 * plaintext operand (an all-digit wire name, circuit.rs:1328-1334) carries it in scalar_lo / scalar_hi.  The call needs
 * helm_host_radix_scratch_rows() rows of the same table from scratch_first_row on (beyond every integer).  What a Rust
 * `impl EvalCircuit<FheType> for HipArithCircuit` forwards each level to. */
typedef enum {
    HELM_RADIX_COPY = 0, HELM_RADIX_ADD, HELM_RADIX_SUB, HELM_RADIX_MUL, HELM_RADIX_DIV, HELM_RADIX_SHL, HELM_RADIX_SHR,
    HELM_RADIX_ADD_SCALAR, HELM_RADIX_SUB_SCALAR, HELM_RADIX_MUL_SCALAR, HELM_RADIX_DIV_SCALAR, HELM_RADIX_SHL_SCALAR,
    HELM_RADIX_SHR_SCALAR
} helm_radix_kind;
typedef struct {
    int32_t kind;      /* helm_radix_kind */
    int32_t a, b, out; /* first rows; b = -1 for copy and the scalar forms */
    uint64_t scalar_lo, scalar_hi;
} helm_radix_op;
int64_t helm_host_radix_scratch_rows(helm_si_ctx *ctx, int32_t blocks, const helm_radix_op *ops, int64_t count);
int helm_host_radix_level(helm_si_ctx *ctx, helm_si_wires *wires, int32_t blocks, const helm_radix_op *ops, int64_t count,
                          int32_t scratch_first_row, int64_t *pbs_out, int64_t *rounds_out);
char *helm_host_si_circuit_log(helm_si_circuit *c);
/* bootstraps of the last evaluate_encrypted, and (arithmetic) the number of batched rounds */
int64_t helm_host_si_circuit_pbs_per_cycle(const helm_si_circuit *c);
int64_t helm_host_si_circuit_pbs_rounds_per_cycle(const helm_si_circuit *c);
/* Same-cycle memo.  LUT mode: as helm_host_gate_circuit_memo_hits (same cycle, same unmodified map).  Arithmetic mode:
 * keyed on the cycle ALONE, as the reference's *_block methods are (src/gates.rs:307-312; tests/gates_test.rs:196-223
 * evaluates other operands in the same cycle and expects the first result; tests/circuit_test.rs:314-474 passes cycles
 * 1..4 to defeat it): pass a new cycle number for new inputs. */
int64_t helm_host_si_circuit_memo_hits(const helm_si_circuit *c);
int helm_host_si_enc_map_new(helm_si_ctx *server_key, int blocks, helm_si_enc_map **out);
void helm_host_si_enc_map_free(helm_si_enc_map *m);
int helm_host_si_enc_map_blocks(const helm_si_enc_map *m);
int helm_host_si_enc_map_row_words(const helm_si_enc_map *m);
int helm_host_si_enc_map_insert(helm_si_enc_map *m, const char *wire, const uint64_t *lwe);
int helm_host_si_enc_map_get(const helm_si_enc_map *m, const char *wire, uint64_t *lwe_out);
int helm_host_si_enc_map_contains_key(const helm_si_enc_map *m, const char *wire);
char *helm_host_si_enc_map_keys(const helm_si_enc_map *m);

#ifdef __cplusplus
}
#endif
#endif
