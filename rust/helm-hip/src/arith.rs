//! `HipArithCircuit`: the arithmetic-mode implementor of HELM's `EvalCircuit` (reference src/circuit.rs:1113-1483).
//! The reference evaluates each gate of a level with one `FheUintN` operator (src/gates.rs:306-702); here a level is
//! ONE call, `helm_host_radix_level`, over a device-resident table in which an `FheUintN` is N/2 consecutive rows
//! (radix blocks of 2 message bits, least significant first, under PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS,
//! src/bin/helm.rs:83).  The same-cycle memo of the `*_block` methods (gates.rs:307-312) is kept: a repeated cycle
//! returns the cached outputs whatever the operands, as tests/gates_test.rs:196-223 expects.
//!
//! NOT COMPILED in this repository's image (no rustc); tests/c/shim_sequence_si.c issues the same calls in the same
//! order from C.  [RECALLED] tfhe-rs 0.4 items to confirm:
//!   tfhe::{ClientKey, FheUint8..FheUint128}; the integer client key's `encrypt_radix(value, num_blocks)` and the
//!   blocks of a `RadixCiphertext` (`.blocks()[i].ct.as_ref()`); `decrypt_radix`.
use crate::{check_host, DeviceWire};
use helm::circuit::{Circuit, EvalCircuit};
use helm::gates::GateType;
use helm::PtxtType;
use helm_hip_sys as sys;
use std::collections::{HashMap, HashSet};

pub struct HipArithCircuit<'a> {
    circuit: Circuit<'a>,
    client_key: tfhe::ClientKey,
    ctx: *mut sys::helm_si_ctx,
    wires: *mut sys::helm_si_wires,
    row_of: HashMap<String, i32>, // first row of each integer
    blocks: i32,                  // radix blocks per integer: bits / 2
    row_words: usize,
    scratch_first_row: i32,
    evaluated_cycle: Option<usize>,
}

fn is_numeric_string(s: &str) -> bool { s.chars().all(|c| c.is_ascii_digit()) } // circuit.rs:100-102

fn blocks_of(ptxt_type: &str) -> i32 {
    match ptxt_type { "u8" => 4, "u16" => 8, "u32" => 16, "u64" => 32, "u128" => 64, _ => unreachable!() }
}

impl<'a> HipArithCircuit<'a> {
    pub fn new(client_key: tfhe::ClientKey, keys: &crate::keys::StandardKeys64, circuit: Circuit<'a>, device_id: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        crate::check(unsafe { sys::helm_si_ctx_create(device_id, &keys.params, &mut ctx) });
        crate::check(unsafe { sys::helm_si_load_bootstrap_key(ctx, keys.bsk.as_ptr(), keys.bsk.len()) });
        crate::check(unsafe { sys::helm_si_load_keyswitch_key(ctx, keys.ksk.as_ptr(), keys.ksk.len()) });
        let row_words = (keys.params.k * keys.params.N) as usize + 1;
        HipArithCircuit { circuit, client_key, ctx, wires: std::ptr::null_mut(), row_of: HashMap::new(), blocks: 0,
                          row_words, scratch_first_row: 0, evaluated_cycle: None }
    }

    /// The operators of one level as the engine's structs: an all-digit operand is a plaintext scalar
    /// (circuit.rs:1328-1387: ct (op) scalar whatever the operand order), otherwise ct (op) ct (:1389-1435, default arm
    /// = multiplication).  The match on `GateType` is explicit: nothing relies on discriminant order.
    fn level_ops(&self, gates: &[helm::gates::Gate], bits: u32) -> Vec<sys::helm_radix_op> {
        gates.iter().map(|g| {
            let ins = g.get_input_wires();
            let out = self.row_of[&g.get_output_wire()];
            if let Some(scalar_wire) = ins.iter().find(|w| is_numeric_string(w)) {
                let ct = ins.iter().find(|w| !is_numeric_string(w)).expect("Empty ctxt operand!");
                let v: u128 = scalar_wire.parse::<u128>().ok().filter(|v| bits == 128 || *v >> bits == 0).unwrap_or(0);
                let kind = match g.get_gate_type() {
                    GateType::Add => sys::HELM_RADIX_ADD_SCALAR, GateType::Sub => sys::HELM_RADIX_SUB_SCALAR,
                    GateType::Mult => sys::HELM_RADIX_MUL_SCALAR, GateType::Div => sys::HELM_RADIX_DIV_SCALAR,
                    GateType::Shl => sys::HELM_RADIX_SHL_SCALAR, GateType::Shr => sys::HELM_RADIX_SHR_SCALAR,
                    _ => unreachable!(),
                };
                sys::helm_radix_op { kind, a: self.row_of[ct], b: -1, out, scalar_lo: v as u64, scalar_hi: (v >> 64) as u64 }
            } else {
                let kind = match g.get_gate_type() {
                    GateType::Copy => sys::HELM_RADIX_COPY, GateType::Add => sys::HELM_RADIX_ADD,
                    GateType::Sub => sys::HELM_RADIX_SUB, GateType::Div => sys::HELM_RADIX_DIV,
                    GateType::Shl => sys::HELM_RADIX_SHL, GateType::Shr => sys::HELM_RADIX_SHR,
                    _ => sys::HELM_RADIX_MUL, // circuit.rs:1429-1435
                };
                let b = if kind == sys::HELM_RADIX_COPY { -1 } else { self.row_of[&ins[1]] };
                sys::helm_radix_op { kind, a: self.row_of[&ins[0]], b, out, scalar_lo: 0, scalar_hi: 0 }
            }
        }).collect()
    }
}

impl<'a> EvalCircuit<DeviceWire> for HipArithCircuit<'a> {
    /// circuit.rs:1114-1192: inputs <- FheUintN::try_encrypt(value); gate outputs start as FheType::None (rows reserved)
    fn encrypt_inputs(&mut self, wire_set: &HashSet<String>, input_wire_map: &HashMap<String, PtxtType>)
        -> HashMap<String, DeviceWire> {
        let ptxt_type = match input_wire_map.values().next().unwrap() {
            PtxtType::U8(_) => "u8", PtxtType::U16(_) => "u16", PtxtType::U32(_) => "u32", PtxtType::U64(_) => "u64",
            PtxtType::U128(_) => "u128", _ => unreachable!(),
        };
        self.blocks = blocks_of(ptxt_type);
        let mut names: Vec<String> = wire_set.iter().cloned().collect();
        for w in self.circuit.input_wires.iter() { if !wire_set.contains(w) { names.push(w.clone()); } }
        names.sort();
        self.row_of = names.iter().enumerate().map(|(i, n)| (n.clone(), i as i32 * self.blocks)).collect();
        // scratch of the widest level, behind every integer
        let named_rows = names.len() as i64 * self.blocks as i64;
        let mut scratch = 0i64;
        for gates in self.circuit.level_map.values() {
            let ops = self.level_ops(gates, 2 * self.blocks as u32);
            scratch = scratch.max(unsafe { sys::helm_host_radix_scratch_rows(self.ctx, self.blocks, ops.as_ptr(), ops.len() as i64) });
        }
        self.scratch_first_row = named_rows as i32;
        crate::check(unsafe { sys::helm_si_wires_alloc(self.ctx, named_rows + scratch, &mut self.wires) });
        for w in self.circuit.input_wires.iter() {
            let v: u128 = if input_wire_map.is_empty() || input_wire_map.contains_key("dummy") { 0 } else {
                match input_wire_map.get(w) {
                    Some(PtxtType::U8(v)) => *v as u128, Some(PtxtType::U16(v)) => *v as u128, Some(PtxtType::U32(v)) => *v as u128,
                    Some(PtxtType::U64(v)) => *v as u128, Some(PtxtType::U128(v)) => *v,
                    None => panic!("\n Input wire \"{}\" not found in input wires!", w), // circuit.rs:1148
                    _ => unreachable!(),
                }
            };
            // [RECALLED] the blocks of FheUintN::try_encrypt(v): block i encrypts (v >> 2 i) & 3 under the big key
            let words = crate::keys::radix_block_words(&self.client_key, v, self.blocks as usize);
            let rows: Vec<i32> = (0..self.blocks).map(|i| self.row_of[w] + i).collect();
            crate::check(unsafe { sys::helm_si_wires_upload(self.ctx, self.wires, rows.as_ptr(), words.as_ptr(), rows.len() as i64) });
        }
        self.evaluated_cycle = None;
        self.row_of.iter().map(|(k, r)| (k.clone(), DeviceWire(*r))).collect()
    }

    fn init_ready(&mut self) -> HashMap<String, DeviceWire> { unimplemented!() } // circuit.rs:1194-1196
    fn evaluate_ready(&mut self, _: &HashMap<String, DeviceWire>, _: &mut HashMap<String, DeviceWire>) { unimplemented!() }

    /// circuit.rs:1299-1454
    fn evaluate_encrypted(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, cycle: usize, ptxt_type: &str)
        -> HashMap<String, DeviceWire> {
        assert!(self.circuit.gates.is_empty());
        assert!(self.circuit.ordered_gates.is_empty());
        assert_eq!(blocks_of(ptxt_type), self.blocks);
        if self.evaluated_cycle == Some(cycle) {
            return enc_wire_map.clone(); // gates.rs:307-312: every gate hands back this cycle's cached output
        }
        let total_levels = self.circuit.level_map.len();
        let mut levels: Vec<_> = self.circuit.level_map.iter().collect();
        levels.sort_by_key(|(l, _)| **l);
        for (level, gates) in levels {
            let ops = self.level_ops(gates, 2 * self.blocks as u32);
            let (mut pbs, mut rounds) = (0i64, 0i64);
            check_host(unsafe { sys::helm_host_radix_level(self.ctx, self.wires, self.blocks, ops.as_ptr(), ops.len() as i64,
                                                           self.scratch_first_row, &mut pbs, &mut rounds) });
            println!("  Evaluated gates in level [{}/{}]", level, total_levels);
        }
        crate::check(unsafe { sys::helm_si_sync(self.ctx) });
        self.evaluated_cycle = Some(cycle);
        enc_wire_map.clone()
    }

    /// circuit.rs:1456-1483
    fn decrypt_outputs(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, _verbose: bool) -> HashMap<String, PtxtType> {
        self.circuit.output_wires.iter().map(|w| {
            let rows: Vec<i32> = (0..self.blocks).map(|i| enc_wire_map[w].0 + i).collect();
            let mut words = vec![0u64; rows.len() * self.row_words];
            crate::check(unsafe { sys::helm_si_wires_download(self.ctx, self.wires, rows.as_ptr(), words.as_mut_ptr(), rows.len() as i64) });
            let v = crate::keys::radix_decrypt(&self.client_key, &words, self.row_words); // block i = bits 2i+1..2i
            let p = match self.blocks { 4 => PtxtType::U8(v as u8), 8 => PtxtType::U16(v as u16), 16 => PtxtType::U32(v as u32),
                                        32 => PtxtType::U64(v as u64), _ => PtxtType::U128(v) };
            (w.clone(), p)
        }).collect()
    }
}

impl<'a> Drop for HipArithCircuit<'a> {
    fn drop(&mut self) {
        unsafe {
            if !self.wires.is_null() { sys::helm_si_wires_free(self.ctx, self.wires); }
            sys::helm_si_ctx_destroy(self.ctx);
        }
    }
}
