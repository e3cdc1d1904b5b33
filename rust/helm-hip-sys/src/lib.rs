//! `extern "C"` view of include/helm_hip.h, include/helm_shortint.h and the key-import helpers of
//! include/helm_client.h.  One declaration per exported symbol the shim uses; field order and widths
//! follow the headers (checked there by tests/test_abi.py against the built libraries).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct helm_hip_params {
    pub torus_bits: i32,
    pub n: i32,
    pub k: i32,
    pub N: i32,
    pub pbs_l: i32,
    pub pbs_logB: i32,
    pub ks_l: i32,
    pub ks_logB: i32,
    pub pbs_order: i32,       // 0 = bootstrap then keyswitch (tfhe boolean)
    pub grouping_factor: i32, // 1
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct helm_si_params {
    pub n: i32,
    pub k: i32,
    pub N: i32,
    pub pbs_l: i32,
    pub pbs_logB: i32,
    pub ks_l: i32,
    pub ks_logB: i32,
    pub message_modulus: i32,
    pub carry_modulus: i32,
    pub grouping_factor: i32, // 0 | 1 classical, 2 | 3 multi-bit
}

#[repr(C)] pub struct helm_hip_ctx { _p: [u8; 0] }
#[repr(C)] pub struct helm_hip_wires { _p: [u8; 0] }
#[repr(C)] pub struct helm_hip_program { _p: [u8; 0] }
#[repr(C)] pub struct helm_si_ctx { _p: [u8; 0] }
#[repr(C)] pub struct helm_si_wires { _p: [u8; 0] }
#[repr(C)] pub struct helm_comm { _p: [u8; 0] }

/// include/helm_comm.h: bytes of an RCCL unique id (= NCCL_UNIQUE_ID_BYTES)
pub const HELM_COMM_ID_BYTES: usize = 128;

// include/helm_hip.h `helm_gate_op` (declaration order of GateType, reference src/gates.rs:23-45; the shim maps the
// enum with an explicit match, helm-hip/src/lib.rs::gate_opcode)
pub const HELM_GATE_AND: i32 = 0;
pub const HELM_GATE_DFF: i32 = 1;
pub const HELM_GATE_LUT: i32 = 2;
pub const HELM_GATE_MUX: i32 = 3;
pub const HELM_GATE_NAND: i32 = 4;
pub const HELM_GATE_NOR: i32 = 5;
pub const HELM_GATE_NOT: i32 = 6;
pub const HELM_GATE_OR: i32 = 7;
pub const HELM_GATE_XNOR: i32 = 8;
pub const HELM_GATE_XOR: i32 = 9;
pub const HELM_GATE_BUF: i32 = 10;
pub const HELM_GATE_CONST_ONE: i32 = 11;
pub const HELM_GATE_CONST_ZERO: i32 = 12;

// include/helm_host.h `helm_radix_kind` / `helm_radix_op`: one FheUintN operator of a level (gates.rs:306-702)
pub const HELM_RADIX_COPY: i32 = 0;
pub const HELM_RADIX_ADD: i32 = 1;
pub const HELM_RADIX_SUB: i32 = 2;
pub const HELM_RADIX_MUL: i32 = 3;
pub const HELM_RADIX_DIV: i32 = 4;
pub const HELM_RADIX_SHL: i32 = 5;
pub const HELM_RADIX_SHR: i32 = 6;
pub const HELM_RADIX_ADD_SCALAR: i32 = 7;
pub const HELM_RADIX_SUB_SCALAR: i32 = 8;
pub const HELM_RADIX_MUL_SCALAR: i32 = 9;
pub const HELM_RADIX_DIV_SCALAR: i32 = 10;
pub const HELM_RADIX_SHL_SCALAR: i32 = 11;
pub const HELM_RADIX_SHR_SCALAR: i32 = 12;
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct helm_radix_op {
    pub kind: i32,
    pub a: i32,
    pub b: i32,
    pub out: i32,
    pub scalar_lo: u64,
    pub scalar_hi: u64,
}

#[repr(C)] pub struct helm_netlist { _p: [u8; 0] }
#[repr(C)] pub struct helm_circuit { _p: [u8; 0] }
#[repr(C)] pub struct helm_si_circuit { _p: [u8; 0] }
#[repr(C)] pub struct helm_si_enc_map { _p: [u8; 0] }
#[repr(C)] pub struct helm_si_client_key { _p: [u8; 0] } // include/helm_client.h: the library's own client key (tests, benches)

extern "C" {
    // ---- include/helm_hip.h ---------------------------------------------------------------
    pub fn helm_hip_last_error() -> *const c_char;
    pub fn helm_hip_device_count() -> c_int;
    pub fn helm_hip_runtime_copies(paths: *mut c_char, cap: usize) -> c_int;
    pub fn helm_hip_ctx_create(device_id: c_int, params: *const helm_hip_params, out: *mut *mut helm_hip_ctx) -> c_int;
    pub fn helm_hip_ctx_destroy(ctx: *mut helm_hip_ctx) -> c_int;
    pub fn helm_hip_set_stream(ctx: *mut helm_hip_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn helm_hip_sync(ctx: *mut helm_hip_ctx) -> c_int;
    pub fn helm_hip_launch_quantum(ctx: *const helm_hip_ctx) -> i64;
    pub fn helm_hip_launch_costs(ctx: *const helm_hip_ctx, cost: *mut f64) -> c_int;
    pub fn helm_hip_short_root_stages(ctx: *const helm_hip_ctx) -> c_int;
    pub fn helm_hip_field_bits(ctx: *const helm_hip_ctx) -> c_int;
    pub fn helm_hip_load_bootstrap_key(ctx: *mut helm_hip_ctx, bsk_std: *const u32, n_words: usize) -> c_int;
    pub fn helm_hip_load_keyswitch_key(ctx: *mut helm_hip_ctx, ksk: *const u32, n_words: usize) -> c_int;
    pub fn helm_hip_wires_alloc(ctx: *mut helm_hip_ctx, n_wires: i64, out: *mut *mut helm_hip_wires) -> c_int;
    pub fn helm_hip_wires_free(ctx: *mut helm_hip_ctx, w: *mut helm_hip_wires) -> c_int;
    pub fn helm_hip_wires_upload(ctx: *mut helm_hip_ctx, w: *mut helm_hip_wires, idx: *const i32, lwe_host: *const u32, count: i64) -> c_int;
    pub fn helm_hip_wires_download(ctx: *mut helm_hip_ctx, w: *mut helm_hip_wires, idx: *const i32, lwe_host: *mut u32, count: i64) -> c_int;
    pub fn helm_hip_wires_set_trivial(ctx: *mut helm_hip_ctx, w: *mut helm_hip_wires, idx: *const i32, value: *const u8, count: i64) -> c_int;
    pub fn helm_hip_wires_copy(ctx: *mut helm_hip_ctx, src: *mut helm_hip_wires, src_idx: *const i32, dst: *mut helm_hip_wires, dst_idx: *const i32, count: i64) -> c_int;
    pub fn helm_hip_eval_gate_level(ctx: *mut helm_hip_ctx, w: *mut helm_hip_wires, opcode: *const i32, in0: *const i32,
                                    in1: *const i32, in2: *const i32, out: *const i32, count: i64) -> c_int;
    pub fn helm_hip_program_create(ctx: *mut helm_hip_ctx, opcode: *const i32, in0: *const i32, in1: *const i32, in2: *const i32,
                                   out: *const i32, level_offsets: *const i64, n_levels: i64, prog: *mut *mut helm_hip_program) -> c_int;
    pub fn helm_hip_program_run(ctx: *mut helm_hip_ctx, prog: *mut helm_hip_program, w: *mut helm_hip_wires, level_begin: i64, level_end: i64) -> c_int;
    pub fn helm_hip_program_destroy(ctx: *mut helm_hip_ctx, prog: *mut helm_hip_program) -> c_int;
    pub fn helm_hip_program_level_pbs(prog: *mut helm_hip_program, level: i64) -> i64;
    // the whole sharded pass; `exchange` = ncclAllGather(stage, gather, rows * (n + 1) * 4 bytes) on the context's stream
    pub fn helm_hip_program_run_sharded(ctx: *mut helm_hip_ctx, prog: *mut helm_hip_program, w: *mut helm_hip_wires, rank: c_int,
                                        world: c_int, replicate_below: i64, stage_dev: *mut c_void, gather_dev: *mut c_void,
                                        capacity_rows: i64,
                                        exchange: extern "C" fn(*mut c_void, *mut c_void, *mut c_void, i64) -> c_int,
                                        user: *mut c_void) -> c_int;

    // the same pass with the collective inside the library: ncclAllGather through a communicator of include/helm_comm.h
    pub fn helm_hip_program_run_sharded_comm(ctx: *mut helm_hip_ctx, prog: *mut helm_hip_program, w: *mut helm_hip_wires,
                                             comm: *mut helm_comm, replicate_below: i64, overlap: c_int) -> c_int;
    pub fn helm_hip_program_overlap_applies(prog: *mut helm_hip_program) -> c_int;
    pub fn helm_hip_program_chunk_bounds(prog: *mut helm_hip_program, level: i64, world: c_int, bounds: *mut i64) -> c_int;

    // ---- include/helm_comm.h: the library's own RCCL communicator (one process per GPU) --------
    pub fn helm_comm_available() -> c_int;
    pub fn helm_comm_precheck(device_id: c_int) -> c_int;
    pub fn helm_comm_get_unique_id(id: *mut u8) -> c_int;
    pub fn helm_comm_create(device_id: c_int, id: *const u8, rank: c_int, world: c_int, out: *mut *mut helm_comm) -> c_int;
    // a communicator over a transport the host brings (MPI, ...) instead of RCCL
    pub fn helm_comm_create_with_transport(device_id: c_int, rank: c_int, world: c_int,
                                           all_gather: extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize, *mut c_void) -> c_int,
                                           user: *mut c_void, out: *mut *mut helm_comm) -> c_int;
    // ranks as threads of one process: device-to-device copies inside the library instead of RCCL
    pub fn helm_comm_create_in_process(device_ids: *const c_int, world: c_int, timeout_s: f64, out: *mut *mut helm_comm) -> c_int;
    pub fn helm_comm_abort_group(comm: *mut helm_comm) -> c_int;
    pub fn helm_comm_destroy(comm: *mut helm_comm) -> c_int;
    pub fn helm_comm_info(comm: *const helm_comm, rank: *mut c_int, world: *mut c_int, device: *mut c_int,
                          rccl_version: *mut c_int) -> c_int;
    pub fn helm_comm_all_reduce_f64(comm: *mut helm_comm, value: *mut f64, op: c_int) -> c_int;
    pub fn helm_comm_barrier(comm: *mut helm_comm) -> c_int;

    // ---- include/helm_host.h: launch packing of the level map -------------------------------
    pub fn helm_host_pack_levels(opcode: *const i32, in0: *const i32, in1: *const i32, in2: *const i32, out: *const i32,
                                 level_offsets: *const i64, n_levels: i64, quantum: i64, order: *mut i64,
                                 new_offsets: *mut i64, n_launches: *mut i64) -> c_int;

    pub fn helm_host_pack_levels_costed(opcode: *const i32, in0: *const i32, in1: *const i32, in2: *const i32, out: *const i32,
                                        level_offsets: *const i64, n_levels: i64, quantum: i64, quarter_cost: *const f64,
                                        order: *mut i64, new_offsets: *mut i64, n_launches: *mut i64) -> c_int;

    pub fn helm_host_last_error() -> *const c_char;
    // one level of arithmetic-mode operators (FheUintN + - * / << >> and their scalar forms, copy) as one batched call
    pub fn helm_host_radix_scratch_rows(ctx: *mut helm_si_ctx, blocks: i32, ops: *const helm_radix_op, count: i64) -> i64;
    pub fn helm_host_radix_level(ctx: *mut helm_si_ctx, wires: *mut helm_si_wires, blocks: i32, ops: *const helm_radix_op,
                                 count: i64, scratch_first_row: i32, pbs_out: *mut i64, rounds_out: *mut i64) -> c_int;

    // whole-circuit arithmetic / LUT evaluation inside the host library (merged rounds, carry-save planning): the netlist is
    // parsed by the library from the same file; client_key = NULL makes an evaluation-only circuit
    pub fn helm_host_read_verilog_file(file_name: *const c_char, is_arith: c_int, out: *mut *mut helm_netlist) -> c_int;
    pub fn helm_host_netlist_free(nl: *mut helm_netlist);
    pub fn helm_host_netlist_list(nl: *const helm_netlist, which: c_int) -> *mut c_char; // 2 inputs, 3 outputs, 4 dff outputs
    pub fn helm_host_circuit_new(gates_from: *const helm_netlist, input_wires: *const c_char, output_wires: *const c_char,
                                 dff_outputs: *const c_char, out: *mut *mut helm_circuit) -> c_int;
    pub fn helm_host_circuit_free(c: *mut helm_circuit);
    pub fn helm_host_circuit_sort_circuit(c: *mut helm_circuit) -> c_int;
    pub fn helm_host_circuit_compute_levels(c: *mut helm_circuit) -> c_int;
    pub fn helm_host_si_circuit_new(mode: c_int, client_key: *mut helm_si_client_key, server_key: *mut helm_si_ctx, circuit: *const helm_circuit,
                                    out: *mut *mut helm_si_circuit) -> c_int;
    pub fn helm_host_si_circuit_free(c: *mut helm_si_circuit);
    pub fn helm_host_si_circuit_evaluate_encrypted(c: *mut helm_si_circuit, enc_wire_map: *const helm_si_enc_map,
                                                   current_cycle: i64, ptxt_type: *const c_char,
                                                   out: *mut *mut helm_si_enc_map) -> c_int;
    pub fn helm_host_si_enc_map_new(server_key: *mut helm_si_ctx, blocks: c_int, out: *mut *mut helm_si_enc_map) -> c_int;
    pub fn helm_host_si_enc_map_free(m: *mut helm_si_enc_map);
    pub fn helm_host_si_enc_map_insert(m: *mut helm_si_enc_map, wire: *const c_char, lwe: *const u64) -> c_int;
    pub fn helm_host_si_enc_map_get(m: *const helm_si_enc_map, wire: *const c_char, lwe_out: *mut u64) -> c_int;
    pub fn helm_host_free(text: *mut c_char); // strings the library returns

    // ---- include/helm_client.h: key import ---------------------------------------------------
    pub fn helm_keys_last_error() -> *const c_char;
    pub fn helm_keys_bsk32_from_tfhe(p: *const helm_hip_params, tfhe: *const u32, abi: *mut u32, n_words: usize) -> c_int;
    pub fn helm_keys_ksk32_from_tfhe(p: *const helm_hip_params, tfhe: *const u32, abi: *mut u32, n_words: usize) -> c_int;
    pub fn helm_keys_bsk64_from_tfhe(p: *const helm_si_params, tfhe: *const u64, abi: *mut u64, n_words: usize) -> c_int;
    pub fn helm_keys_ksk64_from_tfhe(p: *const helm_si_params, tfhe: *const u64, abi: *mut u64, n_words: usize) -> c_int;

    // ---- include/helm_shortint.h (LUT / arithmetic modes) -------------------------------------
    pub fn helm_si_ctx_create(device_id: c_int, params: *const helm_si_params, out: *mut *mut helm_si_ctx) -> c_int;
    pub fn helm_si_ctx_destroy(ctx: *mut helm_si_ctx) -> c_int;
    pub fn helm_si_ctx_fork(primary: *mut helm_si_ctx, lane_out: *mut *mut helm_si_ctx) -> c_int;
    pub fn helm_si_load_bootstrap_key(ctx: *mut helm_si_ctx, bsk_std: *const u64, n_words: usize) -> c_int;
    pub fn helm_si_load_keyswitch_key(ctx: *mut helm_si_ctx, ksk: *const u64, n_words: usize) -> c_int;
    pub fn helm_si_wires_alloc(ctx: *mut helm_si_ctx, n_rows: i64, out: *mut *mut helm_si_wires) -> c_int;
    pub fn helm_si_wires_free(ctx: *mut helm_si_ctx, w: *mut helm_si_wires) -> c_int;
    pub fn helm_si_wires_upload(ctx: *mut helm_si_ctx, w: *mut helm_si_wires, idx: *const i32, lwe: *const u64, count: i64) -> c_int;
    pub fn helm_si_wires_download(ctx: *mut helm_si_ctx, w: *mut helm_si_wires, idx: *const i32, lwe: *mut u64, count: i64) -> c_int;
    pub fn helm_si_wires_set_trivial(ctx: *mut helm_si_ctx, w: *mut helm_si_wires, idx: *const i32, value: *const u64, count: i64) -> c_int;
    pub fn helm_si_eval_lut_level(ctx: *mut helm_si_ctx, w: *mut helm_si_wires, arity: *const i32, in_idx: *const i32, max_in: c_int,
                                  table: *const u64, out_idx: *const i32, count: i64) -> c_int;
    pub fn helm_si_sync(ctx: *mut helm_si_ctx) -> c_int;
    pub fn helm_si_set_priority(ctx: *mut helm_si_ctx, high: c_int) -> c_int;
    pub fn helm_si_round_capacity(ctx: *mut helm_si_ctx) -> i64;
    // multi-GPU: every bootstrap batch of at least min_batch ciphertexts sharded over the communicator's ranks
    pub fn helm_si_set_exchange_comm(ctx: *mut helm_si_ctx, comm: *mut helm_comm, min_batch: i64, capacity_rows: i64) -> c_int;

    // ---- include/helm_wopbs.h (WoP-PBS wide-LUT path) ------------------------------------------
    pub fn helm_wop_ctx_create(pbs_side: *mut helm_si_ctx, params: *const helm_wop_params, out: *mut *mut helm_wop_ctx) -> c_int;
    pub fn helm_wop_ctx_destroy(ctx: *mut helm_wop_ctx) -> c_int;
    pub fn helm_wop_load_key(ctx: *mut helm_wop_ctx, which: c_int, words: *const u64, n_words: usize, l: i32, log_b: i32) -> c_int;
    pub fn helm_wop_table_words(params: *const helm_wop_params, total_bits: i32) -> usize;
    pub fn helm_wop_make_table(params: *const helm_wop_params, n_blocks: i32, bits_per_block: i32, truth: *const u64,
                               truth_len: usize, table_out: *mut u64) -> c_int;
    pub fn helm_wop_eval_luts(ctx: *mut helm_wop_ctx, w: *mut helm_si_wires, in_idx: *const i32, n_inputs: i32,
                              bits_per_block: i32, tables: *const u64, out_idx: *const i32, count: i64) -> c_int;
    pub fn helm_keys_levels64_reverse(blocks: usize, levels: i32, row_words: usize, src: *const u64, dst: *mut u64,
                                      n_words: usize) -> c_int;
}

/// include/helm_wopbs.h: tfhe::shortint::WopbsParameters, runtime values
#[repr(C)]
#[derive(Clone, Copy, Debug)]
#[allow(non_snake_case)]
pub struct helm_wop_params {
    pub n: i32, pub k: i32, pub N: i32,
    pub pbs_l: i32, pub pbs_logB: i32,
    pub ks_l: i32, pub ks_logB: i32,
    pub pfks_l: i32, pub pfks_logB: i32,
    pub cbs_l: i32, pub cbs_logB: i32,
    pub message_modulus: i32, pub carry_modulus: i32,
}
#[repr(C)]
pub struct helm_wop_ctx { _private: [u8; 0] }
pub const HELM_WOP_KEY_BSK: c_int = 0;
pub const HELM_WOP_KEY_KSK: c_int = 1;
pub const HELM_WOP_KEY_KSK_TO_WOPBS: c_int = 2;
pub const HELM_WOP_KEY_KSK_TO_PBS: c_int = 3;
pub const HELM_WOP_KEY_PFPKSK: c_int = 4;
