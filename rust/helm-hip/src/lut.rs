//! `HipLutCircuit`: the LUT-mode implementor of HELM's `EvalCircuit` (reference src/circuit.rs:969-1111) over
//! include/helm_shortint.h.  `gates::lut()` (src/gates.rs:754-785: pack the input bits into one block, generate the
//! look-up table, keyswitch + programmable bootstrap) is one `helm_si_eval_lut_level` call per netlist level; the
//! `HashMap<String, CtxtShortInt>` values are rows of a device-resident table of big-LWE ciphertexts.
//!
//! NOT COMPILED in this repository's image (no rustc); tests/c/shim_sequence_si.c issues the same calls in the same
//! order from C and is run on the GPU by tests/test_shim_sequence.py.  [RECALLED] tfhe-rs 0.4 items to confirm:
//!   tfhe::shortint::{ClientKey, Ciphertext (alias CiphertextBase<KeyswitchBootstrap>)}, `ciphertext.ct` =
//!   `LweCiphertextOwned<u64>` under the BIG key (k N mask words, then the body), `ClientKey::{encrypt, decrypt}`.
use crate::{check, DeviceWire};
use helm::circuit::{Circuit, EvalCircuit};
use helm::gates::GateType;
use helm::PtxtType;
use helm_hip_sys as sys;
use std::collections::{HashMap, HashSet};
use tfhe::shortint::prelude::*;

pub struct HipLutCircuit<'a> {
    circuit: Circuit<'a>,
    client_key: ClientKey,
    pub(crate) ctx: *mut sys::helm_si_ctx,
    wires: *mut sys::helm_si_wires,
    row_of: HashMap<String, i32>,
    row_words: usize, // k N + 1
    /// same-cycle memo (gates.rs:55-59, 288-292): the cycle whose outputs the table holds
    evaluated_cycle: Option<usize>,
}

impl<'a> HipLutCircuit<'a> {
    /// `LutCircuit::new(client_key, server_key, circuit)` (circuit.rs:393-406) with the server key replaced by an engine
    /// context holding the same key material in the standard domain (keys.rs::standard_keys64).
    pub fn new(client_key: ClientKey, keys: &crate::keys::StandardKeys64, circuit: Circuit<'a>, device_id: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::helm_si_ctx_create(device_id, &keys.params, &mut ctx) });
        check(unsafe { sys::helm_si_load_bootstrap_key(ctx, keys.bsk.as_ptr(), keys.bsk.len()) });
        check(unsafe { sys::helm_si_load_keyswitch_key(ctx, keys.ksk.as_ptr(), keys.ksk.len()) });
        let row_words = (keys.params.k * keys.params.N) as usize + 1;
        HipLutCircuit { circuit, client_key, ctx, wires: std::ptr::null_mut(), row_of: HashMap::new(), row_words,
                        evaluated_cycle: None }
    }

    fn upload(&mut self, rows: &[i32], values: &[u64]) {
        let mut words = Vec::with_capacity(rows.len() * self.row_words);
        for v in values {
            // [RECALLED] Ciphertext { ct: LweCiphertextOwned<u64>, .. }: mask words then body
            words.extend_from_slice(self.client_key.encrypt(*v).ct.as_ref());
        }
        check(unsafe { sys::helm_si_wires_upload(self.ctx, self.wires, rows.as_ptr(), words.as_ptr(), rows.len() as i64) });
    }
}

impl<'a> EvalCircuit<DeviceWire> for HipLutCircuit<'a> {
    /// circuit.rs:970-1000: every gate output <- create_trivial(0); inputs <- client_key.encrypt(bit) (0 when no inputs
    /// were given); DFF state <- encrypt(0)
    fn encrypt_inputs(&mut self, wire_set: &HashSet<String>, input_wire_map: &HashMap<String, PtxtType>)
        -> HashMap<String, DeviceWire> {
        let mut names: Vec<String> = wire_set.iter().cloned().collect();
        for w in self.circuit.input_wires.iter() { if !wire_set.contains(w) { names.push(w.clone()); } }
        names.sort();
        self.row_of = names.iter().enumerate().map(|(i, n)| (n.clone(), i as i32)).collect();
        check(unsafe { sys::helm_si_wires_alloc(self.ctx, names.len() as i64, &mut self.wires) });
        let trivial: Vec<i32> = wire_set.iter().map(|w| self.row_of[w]).collect();
        let zeros = vec![0u64; trivial.len()];
        check(unsafe { sys::helm_si_wires_set_trivial(self.ctx, self.wires, trivial.as_ptr(), zeros.as_ptr(), trivial.len() as i64) });
        let (mut rows, mut values) = (vec![], vec![]);
        for w in self.circuit.input_wires.iter() {
            let v = if input_wire_map.is_empty() || input_wire_map.contains_key("dummy") { 0 } else {
                match input_wire_map.get(w) {
                    Some(PtxtType::Bool(b)) => *b as u64,
                    None => panic!("\n Input wire \"{}\" not found in input wires!", w), // circuit.rs:983
                    _ => unreachable!(),
                }
            };
            rows.push(self.row_of[w]);
            values.push(if self.circuit.dff_outputs.contains(w) { 0 } else { v });
        }
        self.upload(&rows, &values);
        self.evaluated_cycle = None;
        self.row_of.iter().map(|(k, r)| (k.clone(), DeviceWire(*r))).collect()
    }

    /// circuit.rs:1002-1010
    fn init_ready(&mut self) -> HashMap<String, DeviceWire> {
        // the READY-latched copies live behind the netlist's rows; allocated with the table in a full implementation
        self.circuit.output_wires.iter().map(|w| (w.clone(), DeviceWire(self.row_of[w]))).collect()
    }

    /// circuit.rs:1012-1030: valid = READY * new + (1 - READY) * valid, here ONE 3-input look-up per output
    /// (index = READY << 2 | new << 1 | valid, table 0xCA) instead of two multiplications
    fn evaluate_ready(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, valid_outputs: &mut HashMap<String, DeviceWire>) {
        let ready = enc_wire_map["READY"].0;
        let n = valid_outputs.len();
        let (arity, table) = (vec![3i32; n], vec![0xCAu64; n]);
        let (mut in_idx, mut out) = (vec![], vec![]);
        for (k, v) in valid_outputs.iter() {
            in_idx.extend_from_slice(&[ready, enc_wire_map[k].0, v.0]);
            out.push(v.0);
        }
        check(unsafe { sys::helm_si_eval_lut_level(self.ctx, self.wires, arity.as_ptr(), in_idx.as_ptr(), 3, table.as_ptr(),
                                                   out.as_ptr(), n as i64) });
    }

    /// circuit.rs:1032-1083: per level, every LUT gate of the level in ONE call (arity 0 = DFF copy, circuit.rs:1067)
    fn evaluate_encrypted(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, cycle: usize, _ptxt_type: &str)
        -> HashMap<String, DeviceWire> {
        assert!(self.circuit.gates.is_empty());
        assert!(self.circuit.ordered_gates.is_empty());
        if self.evaluated_cycle == Some(cycle) {
            return enc_wire_map.clone(); // gates.rs:288-292: the cached outputs of this cycle are in the table
        }
        let total_levels = self.circuit.level_map.len();
        let mut levels: Vec<_> = self.circuit.level_map.iter().collect();
        levels.sort_by_key(|(l, _)| **l);
        for (level, gates) in levels {
            let max_in = gates.iter().map(|g| g.get_input_wires().len()).max().unwrap_or(1).max(1);
            let (mut arity, mut in_idx, mut table, mut out) = (vec![], vec![], vec![], vec![]);
            for g in gates {
                let ins = g.get_input_wires();
                let mut slot = vec![-1i32; max_in];
                for (q, w) in ins.iter().enumerate() { slot[q] = self.row_of[w]; }
                in_idx.extend(slot);
                out.push(self.row_of[&g.get_output_wire()]);
                match g.get_gate_type() {
                    GateType::Lut => {
                        // truth table as a bit mask: bit i = lut_const[i] & 1 (gates.rs:746-752)
                        let bits = g.get_lut_const().as_ref().expect("Lut const not provided").iter().enumerate()
                            .fold(0u64, |m, (i, v)| m | ((*v & 1) << i));
                        arity.push(ins.len() as i32);
                        table.push(bits);
                    }
                    _ => { arity.push(0); table.push(0); } // evaluate_encrypted_dff: clone of the first input
                }
            }
            check(unsafe { sys::helm_si_eval_lut_level(self.ctx, self.wires, arity.as_ptr(), in_idx.as_ptr(), max_in as i32,
                                                       table.as_ptr(), out.as_ptr(), arity.len() as i64) });
            println!("  Evaluated gates in level [{}/{}]", level, total_levels);
        }
        check(unsafe { sys::helm_si_sync(self.ctx) });
        self.evaluated_cycle = Some(cycle);
        enc_wire_map.clone() // rows are stable; the values changed in HBM
    }

    /// circuit.rs:1085-1110
    fn decrypt_outputs(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, _verbose: bool) -> HashMap<String, PtxtType> {
        let names: Vec<&String> = self.circuit.output_wires.iter().collect();
        let idx: Vec<i32> = names.iter().map(|w| enc_wire_map[*w].0).collect();
        let mut words = vec![0u64; idx.len() * self.row_words];
        check(unsafe { sys::helm_si_wires_download(self.ctx, self.wires, idx.as_ptr(), words.as_mut_ptr(), idx.len() as i64) });
        names.iter().zip(words.chunks(self.row_words)).map(|(w, ct)| {
            // [RECALLED] rebuild the shortint ciphertext around the downloaded LWE (degree = message_modulus - 1)
            let ct = crate::keys::shortint_from_words(&self.client_key, ct);
            ((*w).clone(), PtxtType::U64(self.client_key.decrypt(&ct)))
        }).collect()
    }
}

impl<'a> Drop for HipLutCircuit<'a> {
    fn drop(&mut self) {
        unsafe {
            if !self.wires.is_null() { sys::helm_si_wires_free(self.ctx, self.wires); }
            sys::helm_si_ctx_destroy(self.ctx);
        }
    }
}
