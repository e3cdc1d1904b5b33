//! `HipArithCircuitWhole`: arithmetic mode with the WHOLE evaluation inside the host library
//! (`helm_host_si_circuit_evaluate_encrypted` on an evaluation-only circuit, `client_key = NULL`), instead of one
//! `helm_host_radix_level` call per level as `HipArithCircuit` does.  The library then plans across levels: products
//! and sums that feed additions stay in carry-save form, sub-circuits that share no wire run as chains whose look-up
//! rounds are merged into launches of at most the device's capacity - chi-squared u32 takes 20
//! bootstrap launches in a row instead of the 26 rounds of the level-by-level walk.  The netlist is read by the library
//! from the same file HELM's parser reads (same dialect, same level map: tests/test_verilog_parser.py, tests/test_circuit.py);
//! encryption and decryption stay with tfhe's client key, ciphertext words cross through the encrypted map.
//!
//! NOT COMPILED in this repository's image (no rustc); tests/test_gpu_modes.py::test_evaluation_only_circuit_without_a_client_key
//! drives the same C calls.  [RECALLED] tfhe-rs 0.4 items as in arith.rs.
use crate::check_host;
use helm::PtxtType;
use helm_hip_sys as sys;
use std::collections::{HashMap, HashSet};
use std::ffi::{CStr, CString};

pub struct HipArithCircuitWhole {
    client_key: tfhe::ClientKey,
    ctx: *mut sys::helm_si_ctx,
    circuit: *mut sys::helm_circuit,
    evaluator: *mut sys::helm_si_circuit,
    row_words: usize,
}

fn list(nl: *const sys::helm_netlist, which: i32) -> CString {
    let p = unsafe { sys::helm_host_netlist_list(nl, which) };
    let s = unsafe { CStr::from_ptr(p) }.to_owned();
    unsafe { sys::helm_host_free(p) };
    s
}

impl HipArithCircuitWhole {
    /// `netlist`: the structural Verilog file `helm --arithmetic` is given (src/bin/helm.rs:204-222).
    pub fn new(client_key: tfhe::ClientKey, keys: &crate::keys::StandardKeys64, netlist: &str, device_id: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        crate::check(unsafe { sys::helm_si_ctx_create(device_id, &keys.params, &mut ctx) });
        crate::check(unsafe { sys::helm_si_load_bootstrap_key(ctx, keys.bsk.as_ptr(), keys.bsk.len()) });
        crate::check(unsafe { sys::helm_si_load_keyswitch_key(ctx, keys.ksk.as_ptr(), keys.ksk.len()) });
        let path = CString::new(netlist).unwrap();
        let (mut nl, mut circuit, mut evaluator) = (std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut());
        check_host(unsafe { sys::helm_host_read_verilog_file(path.as_ptr(), 1, &mut nl) });
        let (ins, outs, dffs) = (list(nl, 2), list(nl, 3), list(nl, 4));
        check_host(unsafe { sys::helm_host_circuit_new(nl, ins.as_ptr(), outs.as_ptr(), dffs.as_ptr(), &mut circuit) });
        unsafe { sys::helm_host_netlist_free(nl) };
        check_host(unsafe { sys::helm_host_circuit_sort_circuit(circuit) });
        check_host(unsafe { sys::helm_host_circuit_compute_levels(circuit) });
        // client_key = NULL: evaluation only, this side keeps the keys
        check_host(unsafe { sys::helm_host_si_circuit_new(1, std::ptr::null_mut(), ctx, circuit, &mut evaluator) });
        let row_words = (keys.params.k * keys.params.N) as usize + 1;
        HipArithCircuitWhole { client_key, ctx, circuit, evaluator, row_words }
    }

    /// circuit.rs:1113-1191 + 1198-1483 in one call: every wire of `wire_set` goes in (inputs encrypted under tfhe's key,
    /// the others as trivial zeros, as the reference's encrypt_inputs leaves them), every wire comes back.
    pub fn evaluate(&mut self, wire_set: &HashSet<String>, inputs: &HashMap<String, PtxtType>, cycle: usize, ptxt_type: &str)
        -> HashMap<String, Vec<u64>> { // ciphertext words per wire: `blocks` rows of k N + 1 words (keys::radix_decrypt)
        let blocks = match ptxt_type { "u8" => 4, "u16" => 8, "u32" => 16, "u64" => 32, "u128" => 64, _ => unreachable!() };
        let mut map = std::ptr::null_mut();
        check_host(unsafe { sys::helm_host_si_enc_map_new(self.ctx, blocks, &mut map) });
        let zeros = vec![0u64; blocks as usize * self.row_words];
        for wire in wire_set {
            let words = match inputs.get(wire) {
                Some(PtxtType::U8(v)) => crate::keys::radix_block_words(&self.client_key, *v as u128, blocks as usize),
                Some(PtxtType::U16(v)) => crate::keys::radix_block_words(&self.client_key, *v as u128, blocks as usize),
                Some(PtxtType::U32(v)) => crate::keys::radix_block_words(&self.client_key, *v as u128, blocks as usize),
                Some(PtxtType::U64(v)) => crate::keys::radix_block_words(&self.client_key, *v as u128, blocks as usize),
                Some(PtxtType::U128(v)) => crate::keys::radix_block_words(&self.client_key, *v, blocks as usize),
                Some(_) => unreachable!(),
                None => zeros.clone(), // not an input: a trivial encryption of zero (all words zero)
            };
            let name = CString::new(wire.as_str()).unwrap();
            check_host(unsafe { sys::helm_host_si_enc_map_insert(map, name.as_ptr(), words.as_ptr()) });
        }
        let ty = CString::new(ptxt_type).unwrap();
        let mut out = std::ptr::null_mut();
        check_host(unsafe { sys::helm_host_si_circuit_evaluate_encrypted(self.evaluator, map, cycle as i64, ty.as_ptr(), &mut out) });
        let mut result = HashMap::new();
        for wire in wire_set {
            let name = CString::new(wire.as_str()).unwrap();
            let mut words = vec![0u64; blocks as usize * self.row_words];
            check_host(unsafe { sys::helm_host_si_enc_map_get(out, name.as_ptr(), words.as_mut_ptr()) });
            result.insert(wire.clone(), words);
        }
        unsafe { sys::helm_host_si_enc_map_free(map); sys::helm_host_si_enc_map_free(out); }
        result
    }
}

impl Drop for HipArithCircuitWhole {
    fn drop(&mut self) {
        unsafe {
            sys::helm_host_si_circuit_free(self.evaluator);
            sys::helm_host_circuit_free(self.circuit);
            sys::helm_si_ctx_destroy(self.ctx);
        }
    }
}
