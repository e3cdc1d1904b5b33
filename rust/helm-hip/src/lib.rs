//! `HipGateCircuit`: one more implementor of HELM's `EvalCircuit` (reference src/circuit.rs:35-58) next
//! to `GateCircuit` (:449-577) and `CircuitCuda` (:579-967).  Same parser, same `Circuit`, same trait;
//! the per-gate `tfhe::boolean::ServerKey` calls of src/gates.rs:254-275 become one engine call per
//! evaluation over a wire table that lives in HBM.
//!
//! NOT COMPILED in this repository's image (no rustc); tests/c/shim_sequence.c issues the same calls in the
//! same order from C and is run on the GPU by tests/test_shim_sequence.py.
pub mod arith;
pub mod arith_whole;
pub mod keys;
pub mod lut;
pub mod multi_gpu;
pub mod wopbs;

pub use arith::HipArithCircuit;
pub use lut::HipLutCircuit;
pub use multi_gpu::HipComm;

use helm::circuit::{Circuit, EvalCircuit};
use helm::gates::GateType;
use helm::PtxtType;
use helm_hip_sys as sys;
use std::collections::{HashMap, HashSet};
use std::ffi::CStr;
use tfhe::boolean::prelude::*;

/// What `Ciphertext` is in `HashMap<String, Ciphertext>`: a row of the device wire table.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct DeviceWire(pub i32);

pub struct HipGateCircuit<'a> {
    circuit: Circuit<'a>,
    client_key: ClientKey,
    ctx: *mut sys::helm_hip_ctx,
    wires: *mut sys::helm_hip_wires,
    prog: *mut sys::helm_hip_program,
    row_of: HashMap<String, i32>,
    n_launches: i64,
    lwe_words: usize, // n + 1
    // multi-GPU (multi_gpu.rs): a communicator of include/helm_comm.h, or null for one GPU
    pub(crate) comm: *mut sys::helm_comm,
    pub(crate) world: i32,
    pub(crate) replicate_below: i64,
    pub(crate) overlap: bool, // a launch's all-gather + scatter beside the launches that do not need its outputs
}

pub(crate) fn check(rc: i32) {
    if rc != 0 {
        // the engine never aborts: a status + message, turned back into the reference's panic here
        let m = unsafe { CStr::from_ptr(sys::helm_hip_last_error()) };
        panic!("{}", m.to_string_lossy());
    }
}

/// One HIP runtime per process: a host that links its own HIP next to this library must not end up with two
/// copies of libamdhip64 mapped - a stream or device pointer of one is not valid in the other.  Called by every constructor
/// of this crate before the first handle could cross; panics (HELM's own error style) with the paths of the copies.
pub fn assert_one_hip_runtime() {
    let mut buf = vec![0u8; 4096];
    let n = unsafe { sys::helm_hip_runtime_copies(buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len()) };
    if n > 1 {
        let paths = String::from_utf8_lossy(&buf[..buf.iter().position(|&b| b == 0).unwrap_or(0)]).replace('\n', ", ");
        panic!("helm-hip: {} HIP runtimes are mapped into this process ({}): load libhelm_hip.so after the host's own libamdhip64", n, paths);
    }
}

pub(crate) fn check_host(rc: i32) {
    if rc != 0 {
        let m = unsafe { CStr::from_ptr(sys::helm_host_last_error()) };
        panic!("{}", m.to_string_lossy());
    }
}

/// `GateType` -> the engine's opcode, spelled out: nothing relies on the enum's discriminant order
/// (reference src/gates.rs:23-45; include/helm_hip.h `helm_gate_op`).
fn gate_opcode(t: GateType) -> i32 {
    match t {
        GateType::And => sys::HELM_GATE_AND,
        GateType::Dff => sys::HELM_GATE_DFF,
        GateType::Lut => sys::HELM_GATE_LUT,
        GateType::Mux => sys::HELM_GATE_MUX,
        GateType::Nand => sys::HELM_GATE_NAND,
        GateType::Nor => sys::HELM_GATE_NOR,
        GateType::Not => sys::HELM_GATE_NOT,
        GateType::Or => sys::HELM_GATE_OR,
        GateType::Xnor => sys::HELM_GATE_XNOR,
        GateType::Xor => sys::HELM_GATE_XOR,
        GateType::Buf => sys::HELM_GATE_BUF,
        GateType::ConstOne => sys::HELM_GATE_CONST_ONE,
        GateType::ConstZero => sys::HELM_GATE_CONST_ZERO,
        // arithmetic operators cannot appear in a boolean circuit: same panic as gates.rs:257-264
        _ => panic!("Arithmetic gates can't be mixed with Boolean ones!"),
    }
}

impl<'a> HipGateCircuit<'a> {
    /// `GateCircuit::new(client_key, server_key, circuit)` (circuit.rs:383-391) with the server key
    /// replaced by an engine context holding the same key material.
    pub fn new(client_key: ClientKey, std_keys: &keys::StandardKeys, circuit: Circuit<'a>, device_id: i32) -> Self {
        assert_one_hip_runtime();
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::helm_hip_ctx_create(device_id, &std_keys.params, &mut ctx) });
        check(unsafe { sys::helm_hip_load_bootstrap_key(ctx, std_keys.bsk.as_ptr(), std_keys.bsk.len()) });
        check(unsafe { sys::helm_hip_load_keyswitch_key(ctx, std_keys.ksk.as_ptr(), std_keys.ksk.len()) });
        HipGateCircuit {
            circuit, client_key, ctx, wires: std::ptr::null_mut(), prog: std::ptr::null_mut(),
            row_of: HashMap::new(), n_launches: 0, lwe_words: std_keys.params.n as usize + 1,
            comm: std::ptr::null_mut(), world: 1, replicate_below: 256, overlap: false,
        }
    }

    /// Flatten `circuit.level_map` (sorted by level) into index arrays, pack the launches and upload the
    /// program once.  opcode = `gate_opcode(GateType)`; MUX: in2 = select (gates.rs:265).
    fn build_program(&mut self) {
        let (mut op, mut i0, mut i1, mut i2, mut out) = (vec![], vec![], vec![], vec![], vec![]);
        let mut off: Vec<i64> = vec![0];
        let mut levels: Vec<_> = self.circuit.level_map.iter().collect();
        levels.sort_by_key(|(l, _)| **l);
        for (_, gates) in levels {
            for g in gates {
                let ins = g.get_input_wires();
                let row = |i: usize| ins.get(i).map(|w| self.row_of[w]).unwrap_or(-1);
                op.push(gate_opcode(g.get_gate_type()));
                i0.push(row(0)); i1.push(row(1)); i2.push(row(2));
                out.push(self.row_of[&g.get_output_wire()]);
            }
            off.push(op.len() as i64);
        }
        // launch packing: whole lockstep rounds per launch (helm_amd/csrc/host/level_pack.cpp)
        let total = op.len();
        let (mut order, mut poff, mut n_launch) = (vec![0i64; total], vec![0i64; total + 1], 0i64);
        let q = unsafe { sys::helm_hip_launch_quantum(self.ctx) };
        // the engine's cost per launch width (wide / duo / partial / full lockstep round): launches narrower than a round
        // take its most efficient width; sharded over N ranks the quantum is N rounds (multi_gpu.rs)
        let mut quarter_cost = [1.0f64; 4];
        check(unsafe { sys::helm_hip_launch_costs(self.ctx, quarter_cost.as_mut_ptr()) });
        let world = if self.comm.is_null() { 1 } else { self.world as i64 };
        let rc = unsafe { sys::helm_host_pack_levels_costed(op.as_ptr(), i0.as_ptr(), i1.as_ptr(), i2.as_ptr(), out.as_ptr(),
                                                            off.as_ptr(), off.len() as i64 - 1, q * world, quarter_cost.as_ptr(),
                                                            order.as_mut_ptr(), poff.as_mut_ptr(), &mut n_launch) };
        assert!(rc >= 0, "pack_levels failed");
        let pick = |v: &Vec<i32>| order.iter().map(|&g| v[g as usize]).collect::<Vec<i32>>();
        let (op, i0, i1, i2, out) = (pick(&op), pick(&i0), pick(&i1), pick(&i2), pick(&out));
        check(unsafe { sys::helm_hip_program_create(self.ctx, op.as_ptr(), i0.as_ptr(), i1.as_ptr(), i2.as_ptr(), out.as_ptr(),
                                                    poff.as_ptr(), n_launch, &mut self.prog) });
        self.n_launches = n_launch;
    }

    fn ct_words(&self, ct: &Ciphertext) -> Vec<u32> {
        // [RECALLED] tfhe 0.4: `Ciphertext::Encrypted(LweCiphertextOwned<u32>)`, mask words then body
        match ct {
            Ciphertext::Encrypted(lwe) => lwe.as_ref().to_vec(),
            Ciphertext::Trivial(b) => {
                let mut v = vec![0u32; self.lwe_words];
                *v.last_mut().unwrap() = if *b { 1 << 29 } else { 7 << 29 }; // circuit.rs:29,33
                v
            }
        }
    }
}

impl<'a> EvalCircuit<DeviceWire> for HipGateCircuit<'a> {
    /// circuit.rs:450-480: gate outputs <- trivial_encrypt(false); inputs and DFF state <- client_key.encrypt(v)
    fn encrypt_inputs(&mut self, wire_set: &HashSet<String>, input_wire_map: &HashMap<String, PtxtType>)
        -> HashMap<String, DeviceWire> {
        let mut names: Vec<String> = wire_set.iter().cloned().collect();
        for w in self.circuit.input_wires.iter() { if !wire_set.contains(w) { names.push(w.clone()); } }
        names.sort();
        self.row_of = names.iter().enumerate().map(|(i, n)| (n.clone(), i as i32)).collect();
        check(unsafe { sys::helm_hip_wires_alloc(self.ctx, names.len() as i64, &mut self.wires) });

        let trivial: Vec<i32> = wire_set.iter().map(|w| self.row_of[w]).collect();
        let zeros = vec![0u8; trivial.len()];
        check(unsafe { sys::helm_hip_wires_set_trivial(self.ctx, self.wires, trivial.as_ptr(), zeros.as_ptr(), trivial.len() as i64) });

        let (mut idx, mut words) = (vec![], vec![]);
        for w in self.circuit.input_wires.iter() {
            let v = match input_wire_map.get(w) {
                Some(PtxtType::Bool(b)) => *b,
                None if input_wire_map.contains_key("dummy") => false,       // lib.rs:166-178
                _ => panic!("\n Input wire \"{}\" not in input wires!", w),    // circuit.rs:465
            };
            // a DFF output is also an input wire and starts at encrypt(false) (circuit.rs:474-476): one row, once
            let v = if self.circuit.dff_outputs.contains(w) { false } else { v };
            idx.push(self.row_of[w]);
            words.extend(self.ct_words(&self.client_key.encrypt(v)));
        }
        check(unsafe { sys::helm_hip_wires_upload(self.ctx, self.wires, idx.as_ptr(), words.as_ptr(), idx.len() as i64) });
        self.build_program();
        self.row_of.iter().map(|(k, r)| (k.clone(), DeviceWire(*r))).collect()
    }

    /// circuit.rs:506-549: the whole level loop is one call (helm_hip_program_run(l, l + 1) in a loop keeps the
    /// per-level progress lines of :542).
    fn evaluate_encrypted(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, _current_cycle: usize, _ptxt_type: &str)
        -> HashMap<String, DeviceWire> {
        if self.comm.is_null() {
            check(unsafe { sys::helm_hip_program_run(self.ctx, self.prog, self.wires, 0, self.n_launches) });
        } else {
            // launches split over the ranks, outputs all-gathered with ncclAllGather inside the library (multi_gpu.rs)
            check(unsafe { sys::helm_hip_program_run_sharded_comm(self.ctx, self.prog, self.wires, self.comm, self.replicate_below, self.overlap as std::os::raw::c_int) });
        }
        check(unsafe { sys::helm_hip_sync(self.ctx) });
        enc_wire_map.clone() // rows are stable; the values changed in HBM
    }

    /// circuit.rs:482-490
    fn init_ready(&mut self) -> HashMap<String, DeviceWire> {
        self.circuit.output_wires.iter().map(|w| (w.clone(), DeviceWire(self.row_of[w]))).collect()
    }

    /// circuit.rs:492-504: valid = mux(READY, new, valid) per output - one level of MUX gates
    fn evaluate_ready(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, valid_outputs: &mut HashMap<String, DeviceWire>) {
        let ready = enc_wire_map["READY"].0;
        let n = valid_outputs.len();
        let (op, mut i0, mut i1, i2, mut out) = (vec![sys::HELM_GATE_MUX; n], vec![], vec![], vec![ready; n], vec![]);
        for (k, v) in valid_outputs.iter() {
            i0.push(enc_wire_map[k].0); i1.push(v.0); out.push(v.0);
        }
        check(unsafe { sys::helm_hip_eval_gate_level(self.ctx, self.wires, op.as_ptr(), i0.as_ptr(), i1.as_ptr(), i2.as_ptr(),
                                                     out.as_ptr(), n as i64) });
    }

    /// circuit.rs:551-576
    fn decrypt_outputs(&mut self, enc_wire_map: &HashMap<String, DeviceWire>, _verbose: bool) -> HashMap<String, PtxtType> {
        let names: Vec<&String> = self.circuit.output_wires.iter().collect();
        let idx: Vec<i32> = names.iter().map(|w| enc_wire_map[*w].0).collect();
        let mut words = vec![0u32; idx.len() * self.lwe_words];
        check(unsafe { sys::helm_hip_wires_download(self.ctx, self.wires, idx.as_ptr(), words.as_mut_ptr(), idx.len() as i64) });
        names.iter().zip(words.chunks(self.lwe_words)).map(|(w, ct)| {
            // [RECALLED] LweCiphertextOwned::from_container(words, CiphertextModulus::new_native())
            let lwe = tfhe::core_crypto::prelude::LweCiphertextOwned::from_container(
                ct.to_vec(), tfhe::core_crypto::prelude::CiphertextModulus::new_native());
            ((*w).clone(), PtxtType::Bool(self.client_key.decrypt(&Ciphertext::Encrypted(lwe))))
        }).collect()
    }
}

impl<'a> Drop for HipGateCircuit<'a> {
    fn drop(&mut self) {
        unsafe {
            if !self.prog.is_null() { sys::helm_hip_program_destroy(self.ctx, self.prog); }
            if !self.wires.is_null() { sys::helm_hip_wires_free(self.ctx, self.wires); }
            sys::helm_hip_ctx_destroy(self.ctx);
        }
    }
}
