//! `high_precision_lut` on the device: the drop-in for reference src/gates.rs:787-815.
//!
//! The reference builds a radix ciphertext from the gate's input blocks, calls
//! `WopbsKey::keyswitch_to_wopbs_params`, generates the table (`generate_high_precision_lut_radix_helm`,
//! gates.rs:817-864), calls `WopbsKey::wopbs` and `keyswitch_to_pbs_params` and returns block 0.  Here the
//! inputs are rows of the device-resident wire table and the whole sequence is one call,
//! `helm_wop_eval_luts`, batched over every wide gate of a level.
//!
//! NOT COMPILED in this repository's image (no rustc).  [RECALLED] tfhe-rs 0.4 items to confirm:
//!   tfhe::shortint::wopbs::WopbsKey { wopbs_server_key, pbs_server_key, cbs_pfpksk, ksk_pbs_to_wopbs, param }
//!   WopbsParameters { lwe_dimension, glwe_dimension, polynomial_size, pbs_base_log, pbs_level, ks_base_log, ks_level,
//!                     pfks_level, pfks_base_log, cbs_level, cbs_base_log, message_modulus, carry_modulus, .. }
//!   core_crypto::entities::LwePrivateFunctionalPackingKeyswitchKeyList + `.as_ref()` -> &[u64]
//! The wopbs_server_key's bootstrapping key is held in the Fourier domain and has to be regenerated in the
//! standard domain from the WoP-side secret keys, as keys.rs does for the boolean key.
use helm_hip_sys as sys;

pub struct HipWopbsKey {
    ctx: *mut sys::helm_wop_ctx,
    params: sys::helm_wop_params,
    /// bits extracted per block: log2(message_modulus * carry_modulus) is what tfhe's degree bookkeeping extracts
    /// after the cleaning bootstrap (degree = modulus - 1); LUT mode's wires hold one bit, so 1 is enough
    pub bits_per_block: i32,
}

/// Standard-domain words of the five keys, each already in this ABI's order (levels first-to-last).
pub struct WopbsKeyWords<'a> {
    pub bsk: &'a [u64],
    pub ksk: &'a [u64],
    pub ksk_pbs_to_wopbs: &'a [u64],
    pub ksk_wopbs_to_pbs: &'a [u64],
    pub pfpksk: &'a [u64],
    /// decomposition of the two keys between the parameter sets (the PBS side's ks_level / ks_base_log)
    pub between_l: i32,
    pub between_log_b: i32,
}

fn check(rc: i32, what: &str) {
    if rc != 0 {
        let m = unsafe { std::ffi::CStr::from_ptr(sys::helm_hip_last_error()) };
        panic!("{}: {}", what, m.to_string_lossy());
    }
}

/// tfhe's containers keep the levels of keyswitching-type keys last-to-first ([RECALLED]); this ABI first-to-last.
pub fn levels_from_tfhe(blocks: usize, levels: i32, row_words: usize, tfhe: &[u64]) -> Vec<u64> {
    let mut out = vec![0u64; tfhe.len()];
    check(unsafe { sys::helm_keys_levels64_reverse(blocks, levels, row_words, tfhe.as_ptr(), out.as_mut_ptr(), tfhe.len()) },
          "helm_keys_levels64_reverse");
    out
}

impl HipWopbsKey {
    /// `WopbsKey::new_wopbs_key(&cks, &sks, &params)`: the context sits beside the LUT-mode engine `pbs_side`.
    pub fn new(pbs_side: *mut sys::helm_si_ctx, params: sys::helm_wop_params, keys: &WopbsKeyWords, bits_per_block: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::helm_wop_ctx_create(pbs_side, &params, &mut ctx) }, "helm_wop_ctx_create");
        let load = |which, w: &[u64], l, b| check(unsafe { sys::helm_wop_load_key(ctx, which, w.as_ptr(), w.len(), l, b) }, "helm_wop_load_key");
        load(sys::HELM_WOP_KEY_BSK, keys.bsk, 0, 0);
        load(sys::HELM_WOP_KEY_KSK, keys.ksk, 0, 0);
        load(sys::HELM_WOP_KEY_KSK_TO_WOPBS, keys.ksk_pbs_to_wopbs, keys.between_l, keys.between_log_b);
        load(sys::HELM_WOP_KEY_KSK_TO_PBS, keys.ksk_wopbs_to_pbs, keys.between_l, keys.between_log_b);
        load(sys::HELM_WOP_KEY_PFPKSK, keys.pfpksk, 0, 0);
        HipWopbsKey { ctx, params, bits_per_block }
    }

    /// `high_precision_lut(wk_si, wk, sks, lut_const, ctxts)` for every wide gate of a level: `in_rows` holds
    /// n_inputs rows per gate (first = most significant block, gates.rs:795-799), `lut_consts` one table per gate.
    pub fn high_precision_lut_level(&self, wires: *mut sys::helm_si_wires, in_rows: &[i32], n_inputs: i32,
                                    lut_consts: &[&[u64]], out_rows: &[i32]) {
        let words = unsafe { sys::helm_wop_table_words(&self.params, n_inputs * self.bits_per_block) };
        let mut tables = vec![0u64; words * lut_consts.len()];
        for (g, lc) in lut_consts.iter().enumerate() {
            check(unsafe { sys::helm_wop_make_table(&self.params, n_inputs, self.bits_per_block, lc.as_ptr(), lc.len(),
                                                    tables[g * words..].as_mut_ptr()) }, "helm_wop_make_table");
        }
        check(unsafe { sys::helm_wop_eval_luts(self.ctx, wires, in_rows.as_ptr(), n_inputs, self.bits_per_block,
                                               tables.as_ptr(), out_rows.as_ptr(), out_rows.len() as i64) }, "helm_wop_eval_luts");
    }
}

impl Drop for HipWopbsKey {
    fn drop(&mut self) {
        unsafe { sys::helm_wop_ctx_destroy(self.ctx) };
    }
}
